"""Randomised parity fuzz on the GPU box: random index / query shapes, both scan implementations, both rankers,
against the CPU oracle (ids in order, distance bits, rough/precise counters).  Not part of the test suite (run
time grows with ROUNDS);  gpurun -- 'ROUNDS=40 python tests/fuzz_parity.py'."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (test infrastructure: this script is a test driver)
import rabitq_amd as rq  # noqa: E402
from rabitq_amd import _lib, index as ix  # noqa: E402
from tests import synth  # noqa: E402
from tests.test_gpu_parity import _compare_with_oracle  # noqa: E402

_lib.check(_lib.lib().rq_init(0))
rounds = int(os.environ.get("ROUNDS", 30))
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
t0 = time.time()
for it in range(rounds):
    d = int(rng.choice([64, 100, 128, 128, 192, 256, 384, 512, 768, 960]))
    k = int(rng.choice([1, 2, 5, 16, 40, 120, 300]))
    nmax = int(os.environ.get("N_MAX", 12000))
    n = int(rng.integers(max(k, 200), nmax if d <= 256 else max(4000, nmax // 4)))
    nq = int(rng.choice([1, 3, 9, 33, 70, 260, 300, 700]))
    sigma = float(rng.choice([0.4, 0.8, 1.2]))
    x, centres, _ = synth.mixture(n, d, k, sigma=sigma, seed=1000 + it, centre_scale=float(rng.choice([0.3, 0.7, 1.5])))
    if rng.random() < 0.3:   # duplicates and exact centroid copies
        x[: min(30, n)] = x[min(30, n): 2 * min(30, n)][: min(30, n)] if n >= 60 else x[: min(30, n)]
        x[-min(k, 10):] = centres[: min(k, 10)]
    dp = (d + 63) // 64 * 64
    P = synth.random_orthogonal(dp, seed=it) if rng.random() < 0.8 else np.eye(dp, dtype=np.float32)
    oidx = oracle.OracleIndex.build(x, centres, P)
    # engine knobs that must never change a result: raw-vector tiers, chunked scan grids, tile tables
    knobs = {"base_device_mb": int(rng.choice([-1, -1, 0, 1])), "max_scan_blocks": int(rng.choice([0, 0, 3, 40])),
             "scan_tile_table": int(rng.choice([0, 1, 2])), "group_rank": int(rng.choice([0, 1, 2])),
             "rerank_shadow": int(rng.choice([0, 1, 1])), "coarse_impl": int(rng.choice([0, 1, 2])),
             "dense_dir": int(rng.choice([0, 1, 1])), "small_batch": int(rng.choice([0, 0, 1]))}
    for name, v in knobs.items():
        ix.set_option(name, v)
    gidx = rq.RaBitQ.build(x, centres, P)
    queries, _, _ = synth.mixture(nq, d, k, sigma=sigma, seed=5000 + it, centre_scale=0.7)
    if nq > 2:
        queries[1] = x[int(rng.integers(n))]
    impl = int(rng.choice([0, 1, 2]))
    ix.set_option("scan_impl", impl)
    cfgs = []
    for _ in range(2):
        probe = int(rng.choice([1, 2, max(1, k // 2), k, k + 3, 70]))
        topk = int(rng.choice([1, 5, 10, 63, 64, 65, 200]))
        cfgs.append((probe, topk, bool(rng.random() < 0.3)))
    ok = True
    for probe, topk, heur in cfgs:
        try:
            _compare_with_oracle(rq, oracle, oidx, gidx, queries, probe, topk, heur)
        except rq.RabitqError as e:   # the reference panics on the same input (e.g. heuristic ranker with no candidate)
            if e.status != -7:
                raise
        except RuntimeError as e:     # the oracle reports a reference panic for this input: nothing to compare
            if "reference panics" not in str(e):
                raise
    ix.set_option("scan_impl", 0)
    ix.set_option("base_device_mb", -1), ix.set_option("max_scan_blocks", 0), ix.set_option("scan_tile_table", 1)
    ix.set_option("group_rank", 1), ix.set_option("rerank_shadow", 1), ix.set_option("coarse_impl", 0), ix.set_option("dense_dir", 1), ix.set_option("small_batch", 0)
    gidx.close()
    oidx.close()
    print(f"[{it + 1}/{rounds}] n={n} d={d} k={k} nq={nq} impl={impl} knobs={list(knobs.values())} cfgs={cfgs} ok  ({time.time() - t0:.0f}s)", flush=True)
print("fuzz parity: all rounds identical to the oracle")
