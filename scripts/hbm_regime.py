#!/usr/bin/env python3
"""The scan in its HBM-bound regime: small query batches (lists hardly shared) against a full-size index.

Run plain for HIP-event timings, or under rocprofv3 (scripts/pmc_hbm_regime.sh) for the physical HBM bytes
(FETCH_SIZE) of the same launches.  One JSON line on stdout: per batch size the scan time per call (HIP events
on the engine's stream), the algorithmic bytes (sum over probed lists of len * (dim/8 + 16)) and
the number of scan launches per call, so that the PMC summary can attribute dispatches to batch sizes.

    python3 scripts/hbm_regime.py --vectors 100000000 --dim 128 --batches 64,1 --reps 20
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vectors", type=int, default=100_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--lists", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--batches", default="64,1")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sigma", type=float, default=0.5)
    args = ap.parse_args()

    import torch
    import rabitq_amd
    from rabitq_amd import _lib, index as rqi
    from tests import synth

    dev = torch.device("cuda", 0)
    _lib.check(_lib.lib().rq_init(0))
    n, d, k = args.vectors, args.dim, args.lists
    x, centres = synth.device_mixture(n, d, k, args.sigma, dev)
    bmax = max(int(b) for b in args.batches.split(","))
    # (reps + 3 warm-up calls) x the largest batch: every call, warm-up included, sees queries (and so lists) of its own
    queries = synth.device_queries(centres, (args.reps + 3) * bmax, args.sigma, dev)
    idx = rabitq_amd.RaBitQ.build_device(x.data_ptr(), n, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, 99))
    del x
    torch.cuda.empty_cache()
    out_d = torch.empty((bmax, args.topk), device=dev, dtype=torch.float32)
    out_i = torch.zeros((bmax, args.topk), device=dev, dtype=torch.int32)
    out_n = torch.zeros((bmax,), device=dev, dtype=torch.int32)
    rqi.set_profiling(2)
    res = {"config": {"vectors": n, "dim": d, "lists": k, "nprobe": args.nprobe, "topk": args.topk, "reps": args.reps},
           "regimes": []}
    for b in (int(v) for v in args.batches.split(",")):
        def call(q0):
            idx.query_batch_device(queries[q0:q0 + b].data_ptr(), b, d, args.nprobe, args.topk, out_d.data_ptr(),
                                   out_i.data_ptr(), out_n.data_ptr())
        for w in range(3):
            call(w * b)
        acc = {}
        for r in range(args.reps):
            call((3 + r) * b)     # different queries per call: no list stays cache-warm by design
            for key, v in rqi.last_profile().items():
                acc[key] = acc.get(key, 0) + v
        res["regimes"].append({
            "batch": b, "calls": args.reps, "warmup_calls": 3,
            "scan_launches_per_call": acc["scan_launches"] / args.reps,
            "small_batch_passes_per_call": acc.get("small_batch_passes", 0) / args.reps,
            "early_ms_per_call": acc.get("ms_early", 0.0) / args.reps,
            "scan_ms_per_call": acc["ms_scan"] / args.reps, "total_ms_per_call": acc["ms_total"] / args.reps,
            "algorithmic_bytes_per_call": acc["scan_bytes"] / args.reps,
            "algorithmic_GBps": acc["scan_bytes"] / (acc["ms_scan"] * 1e-3) / 1e9,
        })
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
