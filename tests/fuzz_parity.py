"""Randomised parity fuzz on the GPU box: random index / query shapes, data families and scales, both scan
implementations, both rankers, random engine knobs that must never change a result -- against the CPU oracle (ids in
order, distance bits, rough / precise counters).

    gpurun -- 'ROUNDS=400 SEED=3 python tests/fuzz_parity.py'
    (N_MAX = largest index, default 12000; BIG_BATCHES=1 also draws batches of 2100 / 4100 queries, BIG_K=1 also 700 .. 6000 lists)

A seeded, bounded slice of the same rounds runs inside the suite (tests/test_gpu_parity.py::test_fuzz_slice).

Data families (round 3; before that every input was a unit-scale Gaussian mixture):
  gauss       mixture of Gaussians
  sift_u8     integer-valued coordinates 0..255, like SIFT descriptors
  sparse      90 % exact zeros (the strict `> 0` of src/utils.rs:56, zero residual coordinates)
  student_t   heavy tails (Student t, 2.5 degrees of freedom)
  tight       a third of the vectors within 1e-4 of their centroid, some exactly ON it (zero residual: the
              `!norm.is_normal()` branch of src/rabitq.rs:213, huge 1/factor_ip: the matrix-core gate's `safe` cut-off at
              qb + sum_q < 2^19, stage_fill_item, is crossed from both sides)
  near_ties   half of the vectors are noisy copies (1e-6 relative) of twenty rows: rough and exact distances crowd within a
              few ulp of every threshold the rankers hold
Scales 1e-3 .. 3e4 multiply everything (data, centroids, queries).  Query families: mixture draws, exact data rows,
centroids moved by one ulp."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KINDS = ["gauss", "gauss", "sift_u8", "sparse", "student_t", "tight", "near_ties"]
SCALES = [1e-3, 1.0, 1.0, 40.0, 1e3, 3e4]
KNOB_DEFAULTS = {"base_device_mb": -1, "max_scan_blocks": 0, "scan_tile_table": 1, "group_rank": 1, "rerank_shadow": 2,
                 "coarse_impl": 0, "dense_dir": 1, "small_batch": 0, "scan_impl": 0, "small_batch_span": 2560, "stage_growth": 0,
                 "survivor_segments": 1, "scan_gate": 0, "split_rows": 1}


def make_case(rng, it, nmax):
    """One random (base, centroids, P, queries) case + its description."""
    from tests import synth
    d = int(rng.choice([64, 100, 128, 128, 192, 256, 384, 512, 768, 960]))
    k = int(rng.choice([1, 2, 5, 16, 40, 120, 300]))
    if os.environ.get("BIG_K") and rng.random() < 0.25:   # many short lists: the wide probe-selection kernels, ranked groups over thousands of lists
        k = int(rng.choice([700, 2000, 4096, 6000]))
    n_lo, n_hi = max(k, 200), nmax if d <= 256 else max(4000, nmax // 4)
    n = int(rng.integers(n_lo, max(n_hi, n_lo + 200)))   # (BIG_K: more lists than the wide-vector cap on n)
    nq = int(rng.choice([1, 3, 9, 33, 64, 70, 260, 300, 700]))
    if os.environ.get("BIG_BATCHES") and rng.random() < 0.15:   # the large-batch regime (scalar-register coarse kernel, ranked groups, ...)
        nq = int(rng.choice([2100, 4100]))
    sigma = float(rng.choice([0.4, 0.8, 1.2]))
    kind = str(rng.choice(KINDS))
    scale = float(rng.choice(SCALES))
    x, centres, _ = synth.mixture(n, d, k, sigma=sigma, seed=1000 + it, centre_scale=float(rng.choice([0.3, 0.7, 1.5])))
    queries, _, _ = synth.mixture(nq, d, k, sigma=sigma, seed=5000 + it, centre_scale=0.7)
    if kind == "sift_u8":
        x, centres, queries = (np.rint(a * 40 + 128).clip(0, 255) for a in (x, centres, queries))
    elif kind == "sparse":
        x = x * (rng.random(x.shape) < 0.1)
        queries = queries * (rng.random(queries.shape) < 0.1)
        centres = centres * (rng.random(centres.shape) < 0.3)
    elif kind == "student_t":
        x = centres[rng.integers(0, k, n)] + sigma * rng.standard_t(2.5, size=(n, d))
        queries = centres[rng.integers(0, k, nq)] + sigma * rng.standard_t(2.5, size=(nq, d))
    elif kind == "tight":
        m = n // 3
        own = rng.integers(0, k, m)
        x[:m] = centres[own] + 1e-4 * rng.standard_normal((m, d))
        x[: min(m, 3 * k)] = centres[own[: min(m, 3 * k)]]          # exactly on the centroid: zero residual
        queries[: nq // 2] = centres[rng.integers(0, k, nq // 2)] + 1e-3 * rng.standard_normal((nq // 2, d))
    elif kind == "near_ties":
        m = n // 2
        rows = x[: min(20, n)]
        x[:m] = rows[rng.integers(0, len(rows), m)] * (1.0 + 1e-6 * rng.standard_normal((m, d)))
        queries[: nq // 2] = x[rng.integers(0, max(m, 1), nq // 2)] * (1.0 + 1e-6 * rng.standard_normal((nq // 2, d)))
    if rng.random() < 0.3:   # duplicates and exact centroid copies
        x[: min(30, n)] = x[min(30, n): 2 * min(30, n)][: min(30, n)] if n >= 60 else x[: min(30, n)]
        x[-min(k, 10):] = centres[: min(k, 10)]
    x, centres, queries = (np.ascontiguousarray(a * scale, np.float32) for a in (x, centres, queries))
    qkind = str(rng.choice(["mixture", "mixture", "data_rows", "centroid_ulp"]))
    if qkind == "data_rows" and nq > 2:
        queries[: nq // 2] = x[rng.integers(0, n, nq // 2)]
    elif qkind == "centroid_ulp":
        c = centres[rng.integers(0, k, nq)]
        queries = np.nextafter(c, np.where(rng.random(c.shape) < 0.5, np.float32(np.inf), np.float32(-np.inf))).astype(np.float32)
    elif nq > 2:
        queries[1] = x[int(rng.integers(n))]
    dp = (d + 63) // 64 * 64
    P = synth.random_orthogonal(dp, seed=it) if rng.random() < 0.8 else np.eye(dp, dtype=np.float32)
    return x, centres, P, queries, dict(n=n, d=d, k=k, nq=nq, kind=kind, scale=scale, queries=qkind)


def fuzz_round(rq, oracle, rng, it, nmax=12000):
    """One round: build both indexes, compare two (probe, topk, ranker) configurations.  Returns the description and, for
    forced matrix-core rounds, (sub-tile steps, exact-path steps) of the integer gate."""
    from rabitq_amd import index as ix
    from tests.test_gpu_parity import _compare_with_oracle
    x, centres, P, queries, desc = make_case(rng, it, nmax)
    oidx = oracle.OracleIndex.build(x, centres, P)
    # engine knobs that must never change a result
    knobs = {"base_device_mb": int(rng.choice([-1, -1, 0, 1])), "max_scan_blocks": int(rng.choice([0, 0, 3, 40])),
             "scan_tile_table": int(rng.choice([0, 1, 2])), "group_rank": int(rng.choice([0, 1, 2])),
             "rerank_shadow": int(rng.choice([0, 1, 2, 2])), "coarse_impl": int(rng.choice([0, 1, 2, 3, 4])),
             "dense_dir": int(rng.choice([0, 1, 1])), "small_batch": int(rng.choice([0, 0, 1])),
             "small_batch_span": int(rng.choice([100, 2560, 2560, 65536])), "stage_growth": int(rng.choice([0, 0, 2, 16])),
             "scan_impl": int(rng.choice([0, 1, 2])), "survivor_segments": int(rng.choice([1, 1, 2])),
             "scan_gate": int(rng.choice([0, 1, 2])), "split_rows": int(rng.choice([0, 1, 1, 2, 2]))}
    gate = None
    try:
        for name, v in knobs.items():
            ix.set_option(name, v)
        gidx = rq.RaBitQ.build(x, centres, P)
        k = desc["k"]
        cfgs = []
        for _ in range(2):
            probe = int(rng.choice([1, 2, max(1, k // 2), k, k + 3, 70]))
            topk = int(rng.choice([1, 5, 10, 63, 64, 65, 200, 256]))
            cfgs.append((probe, topk, bool(rng.random() < 0.3)))
        if knobs["scan_impl"] == 2:
            ix.set_option("scan_debug", 128)      # count the gate's sub-tile steps and how many took the exact path
            gate = [0, 0]
        for probe, topk, heur in cfgs:
            try:
                _compare_with_oracle(rq, oracle, oidx, gidx, queries, probe, topk, heur)
            except rq.RabitqError as e:   # the reference panics on the same input (e.g. heuristic ranker with no candidate)
                if e.status != -7:
                    raise
            except RuntimeError as e:     # the oracle reports a reference panic for this input: nothing to compare
                if "reference panics" not in str(e):
                    raise
            if gate is not None:
                pr = ix.last_profile()
                gate[0] += pr["matrix_subtile_steps"]
                gate[1] += pr["matrix_exact_steps"]
        gidx.close()
    except BaseException:
        print(f"FUZZ FAILURE round {it}: {desc} knobs={knobs}", flush=True)
        raise
    finally:
        ix.set_option("scan_debug", 0)
        for name, v in KNOB_DEFAULTS.items():
            ix.set_option(name, v)
        oidx.close()
    desc.update(knobs=list(knobs.values()), cfgs=cfgs, gate=gate)
    return desc


def main():
    import oracle  # noqa: E402  (test infrastructure: this script is a test driver)
    import rabitq_amd as rq  # noqa: E402
    from rabitq_amd import _lib
    _lib.check(_lib.lib().rq_init(0))
    rounds = int(os.environ.get("ROUNDS", 30))
    rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
    nmax = int(os.environ.get("N_MAX", 12000))
    t0 = time.time()
    kinds, gates = {}, {"rounds": 0, "with_exact_steps": 0, "all_safe": 0}
    for it in range(rounds):
        desc = fuzz_round(rq, oracle, rng, it, nmax)
        kinds[desc["kind"]] = kinds.get(desc["kind"], 0) + 1
        if desc["gate"] is not None and desc["gate"][0]:
            gates["rounds"] += 1
            gates["with_exact_steps" if desc["gate"][1] else "all_safe"] += 1
        print(f"[{it + 1}/{rounds}] {desc} ok  ({time.time() - t0:.0f}s)", flush=True)
    print(f"fuzz parity: all {rounds} rounds identical to the oracle; families {kinds}; forced matrix-core rounds {gates}")


if __name__ == "__main__":
    main()
