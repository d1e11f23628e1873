"""Benchmark harness with the reference CLI's behaviour (crates/cli/src/main.rs:11-83):

    python -m rabitq_amd.cli -b base.fvecs -c centroids.fvecs -q query.fvecs -t truth.ivecs \
                             -s saved_dir [-p 100] [-k 10] [-h | --heuristic-rank]

* the flags are the reference's, letter for letter: `-h` is the heuristic re-ranker there (`#[argh(switch, short = 'h')]`,
  main.rs:35-37), not help -- argh reserves only `--help` -- so a caller script written for the reference CLI runs unchanged;

* if `--saved` is an existing directory the index is loaded from it, otherwise it is built from
  base + centroids and dumped there (main.rs:52-61);
* every query is issued one at a time and timed individually, QPS = N / sum(elapsed)
  (main.rs:66-80) -- the faithful, latency-bound number; `--batch B` additionally reports the
  batched throughput the GPU engine is designed for;
* prints mean recall@k against the ground-truth ivecs (src/utils.rs:367-379) and the METRICS line
  (src/metrics.rs:30-41).
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np

from . import RaBitQ, calculate_recall, metrics_reset, metrics_str, vecs


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="RaBitQ CLI args", add_help=False)   # `-h` belongs to the reference's switch
    ap.add_argument("--help", action="help", help="display usage information")
    ap.add_argument("-b", "--base", required=True, help="base path")
    ap.add_argument("-c", "--centroids", required=True, help="centroids path")
    ap.add_argument("-q", "--query", required=True, help="query path")
    ap.add_argument("-t", "--truth", required=True, help="truth path")
    ap.add_argument("-p", "--probe", type=int, default=100)
    ap.add_argument("-k", "--topk", type=int, default=10)
    ap.add_argument("-s", "--saved", required=True, help="saved directory")
    ap.add_argument("-h", "--heuristic-rank", "--heuristic_rank", dest="heuristic_rank", action="store_true", help="heuristic re-rank (maybe faster when topk is large)")
    ap.add_argument("--batch", type=int, default=0, help="also time batches of this many queries")
    ap.add_argument("--seed", type=int, default=0, help="seed of the generated rotation when building")
    return ap


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)

    if os.path.isdir(args.saved):
        print(f"loading from {args.saved!r}...")
        index = RaBitQ.load_from_dir(args.saved)
    else:
        print("training...")
        index = RaBitQ.from_path(args.base, args.centroids, seed=args.seed)
        print(f"saving to local file: {args.saved!r}")
        index.dump_to_dir(args.saved)

    queries = vecs.read_vecs(args.query, np.float32)
    truth = vecs.read_vecs(args.truth, np.int32)
    print("querying...")
    metrics_reset()
    total_time, recall = 0.0, 0.0
    for i, q in enumerate(queries):
        t0 = time.perf_counter()
        res = index.query(q, args.probe, args.topk, args.heuristic_rank)
        total_time += time.perf_counter() - t0
        ids = [i_ for _, i_ in res]
        ids += [-1] * (args.topk - len(ids))
        recall += calculate_recall(truth[i], ids, args.topk)
    n = len(queries)
    print(f"QPS: {n / total_time}, recall: {recall / n}")
    print(f"Metrics [{metrics_str()}]")
    if args.batch > 0:
        q = np.stack(queries)
        t0 = time.perf_counter()
        hits = 0
        for s in range(0, n, args.batch):
            _, ids, cnt = index.query_batch(q[s:s + args.batch], args.probe, args.topk, args.heuristic_rank)
            for j in range(ids.shape[0]):
                hits += len(set(ids[j, :cnt[j]].tolist()) & set(truth[s + j][:args.topk].tolist()))
        dt = time.perf_counter() - t0
        print(f"batched ({args.batch}): QPS: {n / dt}, recall: {hits / (n * args.topk)}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
