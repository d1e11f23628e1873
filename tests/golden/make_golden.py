"""Generates tests/golden/*.npz from the CPU oracle (oracle/rabitq_oracle.c).

The reference (Rust, 272 crates.io deps, none vendored) cannot be built or run in this image and
ships no golden vectors of its own, so these fixtures are ORACLE outputs ("parity unpinned by the
reference"); the oracle itself is pinned by tests/test_oracle_kat.py + test_oracle_vs_numpy.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    # name: (n, d, k, sigma, identity_P, nq)
    "d128_k16": (1000, 128, 16, 0.6, False, 8),
    "d100pad_identity": (300, 100, 5, 0.5, True, 6),
    "d768_k4": (96, 768, 4, 0.5, False, 4),
    "d64_k3": (200, 64, 3, 0.7, False, 4),
}
QUERY_CFG = [(4, 10, 0), (64, 3, 0), (2, 20, 1), (64, 10, 1)]  # (probe, topk, heuristic)


def make(name, n, d, k, sigma, ident, nq):
    dpad = (d + 63) // 64 * 64
    x, centres, _ = synth.mixture(n, d, k, sigma=sigma, seed=zlib.crc32(name.encode()) % 1000 + 1, centre_seed=77)
    queries, _, _ = synth.mixture(nq, d, k, sigma=sigma, seed=999, centre_seed=77)
    queries[0] = x[3]  # an exact duplicate of a base vector
    # (`base` in cluster order is base_in[map_ids], zero-padded: not stored)
    P = np.eye(dpad, dtype=np.float32) if ident else synth.random_orthogonal(dpad, seed=5)
    idx = oracle.OracleIndex.build(x, centres, P)
    out = dict(base_in=x, centroids_in=centres, orthogonal=P, queries=queries,
               centroids=idx.centroids, offsets=idx.offsets, map_ids=idx.map_ids,
               codes=idx.codes, factors=idx.factors, rotated=oracle.project_rows(
                   np.pad(x, ((0, 0), (0, dpad - d))), P))
    ys, cl_all, cd_all, prep_lo, prep_delta, prep_sum, prep_planes, rough0 = [], [], [], [], [], [], [], []
    for q in queries:
        y = idx.rotate_query(q)
        ys.append(y)
        cl, cd = idx.coarse_rank(y, k)
        cl_all.append(cl)
        cd_all.append(cd)
        lo, delta, s, planes = idx.query_prep(y, int(cl[0]))
        prep_lo.append(lo), prep_delta.append(delta), prep_sum.append(s), prep_planes.append(planes)
        rough0.append(idx.scan_cluster(int(cl[0]), cd[0], planes, lo, np.float32(s), delta))
    out.update(y=np.array(ys), coarse_cluster=np.array(cl_all), coarse_dist=np.array(cd_all),
               prep_lower=np.array(prep_lo, np.float32), prep_delta=np.array(prep_delta, np.float32),
               prep_sum=np.array(prep_sum, np.uint32), prep_planes=np.array(prep_planes),
               rough_nearest=np.concatenate(rough0), rough_nearest_len=np.array([r.size for r in rough0]))
    for ci, (probe, topk, heur) in enumerate(QUERY_CFG):
        res_id = np.full((nq, topk), 0xFFFFFFFF, np.uint32)
        res_d = np.full((nq, topk), np.nan, np.float32)
        res_n = np.zeros(nq, np.uint32)
        cnt = np.zeros((nq, 2), np.uint64)
        for qi, q in enumerate(queries):
            oracle.metrics_reset()
            dd, ii = idx.query(q, probe, topk, bool(heur))
            res_n[qi] = ii.size
            res_id[qi, :ii.size] = ii
            res_d[qi, :ii.size] = dd
            m = oracle.metrics()
            cnt[qi] = (m["rough"], m["precise"])
        out[f"q{ci}_cfg"] = np.array([probe, topk, heur])
        out[f"q{ci}_ids"], out[f"q{ci}_dist"], out[f"q{ci}_n"], out[f"q{ci}_counts"] = res_id, res_d, res_n, cnt
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    idx.close()


if __name__ == "__main__":
    for name, (n, d, k, sigma, ident, nq) in CASES.items():
        make(name, n, d, k, sigma, ident, nq)
        print("wrote", name)
