"""Self-consistency fuzz AT SCALE (no oracle at these sizes: the engine against itself): one large index, large batches,
random engine knobs that must never change a result -- scan implementation, survivor geometry (uniform / arena), dense
directories, group placement, coarse kernel, stage growth, chunked grids, tile tables, shadow rows off, small-batch path --
against the answer under the default knobs, bit for bit (ids in order, distance bits, counts).

    gpurun -- 'ROUNDS=40 SEED=1 python tests/fuzz_scale.py'          (VECTORS, LISTS, DIM, BATCH, HARD=1, BASE_MB, SHADOW=0 in the environment)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KNOBS = {"max_scan_blocks": [0, 0, 5000, 60000], "scan_tile_table": [0, 1, 2], "group_rank": [0, 1, 2], "coarse_impl": [0, 1, 2, 3, 4],
         "dense_dir": [0, 1, 1], "stage_growth": [0, 0, 2, 4, 16], "scan_impl": [0, 0, 1, 2], "survivor_segments": [0, 1, 2],
         "small_batch": [0, 1], "small_batch_span": [100, 2560, 65536], "scan_gate": [0, 0, 1, 2], "pass_overlap": [0, 1]}
DEFAULTS = {"max_scan_blocks": 0, "scan_tile_table": 1, "group_rank": 1, "coarse_impl": 0, "dense_dir": 1, "stage_growth": 0,
            "scan_impl": 0, "survivor_segments": 1, "small_batch": 0, "small_batch_span": 2560, "scan_gate": 0, "pass_overlap": 1}


def main(**over):
    """`over`: VECTORS / LISTS / DIM / BATCH / ROUNDS / SEED / HARD instead of the environment (the suite's slice)."""
    env = dict(os.environ)
    env.update({k: str(v) for k, v in over.items()})
    import torch
    import rabitq_amd as rq
    from rabitq_amd import _lib, index as ix
    from tests import synth
    _lib.check(_lib.lib().rq_init(0))
    dev = torch.device("cuda", 0)
    n, k, d = int(env.get("VECTORS", 20_000_000)), int(env.get("LISTS", 1024)), int(env.get("DIM", 128))
    rounds, seed = int(env.get("ROUNDS", 20)), int(env.get("SEED", 1))
    hard = bool(env.get("HARD"))
    sigma = 0.5
    rng = np.random.default_rng(seed)
    centres = synth.device_centres(k, d, dev, scale=sigma if hard else 1.0, seed=seed + 10)
    weights = None
    if hard:
        wz = 1.0 / torch.arange(1, k + 1, device=dev, dtype=torch.float64) ** 0.8
        weights = (wz / wz.sum()).float()
    P = synth.random_orthogonal(d, seed=seed + 11)
    chunk = max(262_144, min(4_000_000, (512 << 20) // d))
    # build-time knobs: BASE_MB = HBM budget of the raw vectors in MiB (the rest in pinned host memory), SHADOW=0 = no fp16 rows
    ix.set_option("base_device_mb", int(env.get("BASE_MB", -1)))
    ix.set_option("rerank_shadow", int(env.get("SHADOW", 2)))
    b = rq.RaBitQ.builder(n, d, centres.data_ptr(), k, orthogonal=P)
    for ci, i0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - i0)
        x = synth.device_mixture_chunk(centres, i0, m, sigma, ci, 9000 + seed, 0, k, weights)[0]
        b.assign_chunk(x.data_ptr(), i0, m)
    b.order()
    for ci, i0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - i0)
        x = synth.device_mixture_chunk(centres, i0, m, sigma, ci, 9000 + seed, 0, k, weights)[0].contiguous()
        b.place_chunk(x.data_ptr(), i0, m)
    idx = b.finish()
    ix.set_option("base_device_mb", -1)
    ix.set_option("rerank_shadow", 2)
    del x
    idx_n_host = idx.n - idx.n_hbm
    torch.cuda.empty_cache()
    nq_max = int(env.get("BATCH", 16384))
    queries = synth.device_queries(centres, nq_max, sigma, dev, seed=seed + 12, weights=weights)
    t0 = time.time()

    def run(nq, probe, topk, heur):
        od = torch.full((nq, topk), -1.0, device=dev)
        oi = torch.zeros((nq, topk), device=dev, dtype=torch.int32)
        on = torch.zeros(nq, device=dev, dtype=torch.int32)
        try:
            idx.query_batch_device(queries.data_ptr(), nq, d, probe, topk, od.data_ptr(), oi.data_ptr(), on.data_ptr(), heuristic_rank=heur)
        except rq.RabitqError as e:       # the reference panics on this input (heuristic ranker without a candidate): the same status every time
            if e.status != -7:
                raise
            return None
        return od.cpu().numpy().view(np.uint32), oi.cpu().numpy(), on.cpu().numpy()

    try:
        for it in range(rounds):
            nq = int(rng.choice([48, 300, 2048, nq_max // 2, nq_max]))
            probe = int(rng.choice([8, 32, 64]))
            topk = int(rng.choice([1, 10, 10, 64, 100]))
            heur = bool(rng.random() < 0.2)
            for name, v in DEFAULTS.items():
                ix.set_option(name, v)
            want = run(nq, probe, topk, heur)
            knobs = {name: int(rng.choice(vals)) for name, vals in KNOBS.items()}
            for name, v in knobs.items():
                ix.set_option(name, v)
            got = run(nq, probe, topk, heur)
            got2 = run(nq, probe, topk, heur)     # (capacities / arena learnt by the first call)
            for g in (got, got2):
                if want is None or g is None:
                    if (want is None) != (g is None):
                        print(f"SCALE FUZZ FAILURE round {it}: only one side reports the reference's panic; knobs={knobs}", flush=True)
                        raise SystemExit(1)
                    continue
                if not (np.array_equal(want[2], g[2]) and np.array_equal(want[1], g[1]) and np.array_equal(want[0], g[0])):
                    bad = np.nonzero((want[1] != g[1]).any(axis=1) | (want[2] != g[2]))[0]
                    print(f"SCALE FUZZ FAILURE round {it}: nq={nq} probe={probe} topk={topk} heur={heur} knobs={knobs}; first differing queries {bad[:8]}", flush=True)
                    raise SystemExit(1)
            print(f"[{it + 1}/{rounds}] nq={nq} probe={probe} topk={topk} heur={heur} knobs={list(knobs.values())} ok ({time.time() - t0:.0f}s)", flush=True)
    finally:
        for name, v in DEFAULTS.items():
            ix.set_option(name, v)
        idx.close()
    print(f"scale fuzz: all {rounds} rounds identical to the default-knob answer ({n} x {d}, {k} lists{', hard distribution' if hard else ''}"
          f"{', ' + str(idx_n_host) + ' rows in pinned host memory' if idx_n_host else ''}{', no shadow rows' if env.get('SHADOW') == '0' else ''})")


if __name__ == "__main__":
    main()
