#!/usr/bin/env python3
"""Latency of small batches (1 / 16 / 64 queries) on the BASELINE configs[2] index (100M x 128, 4096 lists, nprobe 64):
device time per call (HIP events around the whole pass), wall time per call, per-kernel-group breakdown, for the few-launch
small-batch path and for the staged path (option small_batch = 1), on fresh queries every call."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vectors", type=int, default=100_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--lists", type=int, default=4096)
    ap.add_argument("--nprobe", type=int, default=64)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--sigma", type=float, default=0.5)
    ap.add_argument("--batches", default="1,16,64")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--spans", default="", help="also try these in-block span caps (small_batch_span), batch 1 and 16")
    args = ap.parse_args()
    import torch
    import rabitq_amd
    from rabitq_amd import _lib, index as rqi
    from tests import synth
    dev = torch.device("cuda", 0)
    _lib.check(_lib.lib().rq_init(0))
    n, d, k = args.vectors, args.dim, args.lists
    centres = synth.device_centres(k, d, dev)
    chunk = max(262_144, min(4_000_000, (512 << 20) // d))
    b = rabitq_amd.RaBitQ.builder(n, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, 99))
    for ci, i0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - i0)
        x = synth.device_mixture_chunk(centres, i0, m, args.sigma, ci)[0]
        b.assign_chunk(x.data_ptr(), i0, m)
    b.order()
    for ci, i0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - i0)
        x = synth.device_mixture_chunk(centres, i0, m, args.sigma, ci)[0].contiguous()
        b.place_chunk(x.data_ptr(), i0, m)
    idx = b.finish()
    del x
    torch.cuda.empty_cache()
    bl = [int(v) for v in args.batches.split(",")]
    queries = synth.device_queries(centres, (args.reps + 3) * max(bl), args.sigma, dev)
    out_d = torch.empty((max(bl), args.topk), device=dev)
    out_i = torch.zeros((max(bl), args.topk), device=dev, dtype=torch.int32)
    out_n = torch.zeros((max(bl),), device=dev, dtype=torch.int32)
    res = {"config": vars(args), "rows": []}
    for small in (0, 1):
        rqi.set_option("small_batch", small)
        for nb in bl:
            row = {"path": "few-launch" if small == 0 else "staged", "batch": nb}
            for level in (2, 1):
                rqi.set_profiling(level)
                acc, wall = {}, 0.0
                for r in range(args.reps + 3):
                    q = queries[r * nb:(r + 1) * nb]
                    t0 = time.perf_counter()
                    idx.query_batch_device(q.data_ptr(), nb, d, args.nprobe, args.topk, out_d.data_ptr(), out_i.data_ptr(), out_n.data_ptr())
                    dt = time.perf_counter() - t0
                    if r >= 3:
                        wall += dt
                        for key, v in rqi.last_profile().items():
                            acc[key] = acc.get(key, 0) + v
                if level == 2:
                    row["device_ms"] = round(acc["ms_total"] / args.reps, 4)
                    row["wall_ms"] = round(wall / args.reps * 1e3, 4)
                    row["scan_ms"] = round(acc["ms_scan"] / args.reps, 4)
                    row["algorithmic_MB"] = round(acc["scan_bytes"] / args.reps / 1e6, 2)
                    row["whole_call_GBps"] = round(acc["scan_bytes"] / (acc["ms_total"] * 1e-3) / 1e9, 1)
                    row["scan_launch_GBps"] = round(acc["scan_bytes"] / (acc["ms_scan"] * 1e-3) / 1e9, 1) if acc["ms_scan"] else None
                    row["rerank_per_query"] = round(acc["rerank_candidates"] / args.reps / nb, 1)
                else:
                    row["breakdown_ms"] = {key[3:]: round(v / args.reps, 4) for key, v in acc.items() if key.startswith("ms_") and v}
            res["rows"].append(row)
            print(json.dumps(row), flush=True)
    rqi.set_option("small_batch", 0)
    for span in [int(v) for v in args.spans.split(",") if v]:
        rqi.set_option("small_batch_span", span)
        rqi.set_profiling(2)
        for nb in (1, 16):
            acc = {}
            for r in range(args.reps + 3):
                q = queries[r * nb:(r + 1) * nb]
                idx.query_batch_device(q.data_ptr(), nb, d, args.nprobe, args.topk, out_d.data_ptr(), out_i.data_ptr(), out_n.data_ptr())
                if r >= 3:
                    for key, v in rqi.last_profile().items():
                        acc[key] = acc.get(key, 0) + v
            print(json.dumps({"span": span, "batch": nb, "device_ms": round(acc["ms_total"] / args.reps, 4),
                              "rerank_per_query": round(acc["rerank_candidates"] / args.reps / nb, 1), "retries": acc["retries"]}), flush=True)
    rqi.set_option("small_batch_span", 2560)
    rqi.set_option("scan_debug", 4096)      # phase stamps of the per-query kernel (stderr), three single queries
    rqi.set_profiling(0)
    for r in range(3):
        q = queries[r:r + 1]
        idx.query_batch_device(q.data_ptr(), 1, d, args.nprobe, args.topk, out_d.data_ptr(), out_i.data_ptr(), out_n.data_ptr())
    rqi.set_option("scan_debug", 0)
    json.dump(res, open("gpurun_out/small_batch_latency.json", "w"), indent=1)


if __name__ == "__main__":
    main()
