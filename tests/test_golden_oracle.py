"""The oracle must reproduce the committed golden fixtures (tests/golden/*.npz) bit for bit.

Fixtures are oracle outputs (generator: tests/golden/make_golden.py); the reference has none of its
own.  The GPU parity tests (test_gpu_*.py) check the HIP path against the same files.
"""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    idx = oracle.OracleIndex.build(g["base_in"], g["centroids_in"], g["orthogonal"])
    for name in ("centroids", "offsets", "map_ids", "codes", "factors"):
        assert np.array_equal(bits(getattr(idx, name)), bits(g[name])), name
    dpad = idx.dim
    padded = np.pad(g["base_in"], ((0, 0), (0, dpad - g["base_in"].shape[1])))
    assert np.array_equal(idx.base, padded[g["map_ids"]])
    for qi, q in enumerate(g["queries"]):
        y = idx.rotate_query(q)
        assert np.array_equal(bits(y), bits(g["y"][qi]))
        cl, cd = idx.coarse_rank(y, idx.k)
        assert np.array_equal(cl, g["coarse_cluster"][qi]) and np.array_equal(bits(cd), bits(g["coarse_dist"][qi]))
    ci = 0
    while f"q{ci}_cfg" in g:
        probe, topk, heur = (int(v) for v in g[f"q{ci}_cfg"])
        for qi, q in enumerate(g["queries"]):
            oracle.metrics_reset()
            d, ids = idx.query(q, probe, topk, bool(heur))
            n = int(g[f"q{ci}_n"][qi])
            assert ids.size == n
            assert np.array_equal(ids, g[f"q{ci}_ids"][qi, :n])
            assert np.array_equal(bits(d), bits(g[f"q{ci}_dist"][qi, :n]))
            m = oracle.metrics()
            assert (m["rough"], m["precise"]) == tuple(int(v) for v in g[f"q{ci}_counts"][qi])
        ci += 1
    assert ci == 4
    idx.close()
