"""Developer tool: time the query pipeline under scan ablations (rq_set_option("scan_debug", bits)).

  bit 0: skip the exact path / emit (no survivors: downstream stages see nothing)
  bit 1: no re-staging of query tiles (every tile re-uses the first one)

Results are wrong under any ablation; only the kernel timings are meaningful.
Run on the GPU box:  gpurun -- 'python scripts/ablate_scan.py'
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rabitq_amd  # noqa: E402
from rabitq_amd import index as rqi  # noqa: E402
from tests import synth  # noqa: E402

n, d, k, nprobe, topk, B = int(os.environ.get("N", 100_000_000)), 128, 4096, 64, 10, int(os.environ.get("B", 10000))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
centres = torch.randn(k, d, generator=g, device=dev)
x = torch.empty((n, d), device=dev)
for ci, i0 in enumerate(range(0, n, 4_000_000)):
    m = min(4_000_000, n - i0)
    g.manual_seed(42 + ci)
    u = torch.randint(0, k, (m,), generator=g, device=dev)
    x[i0:i0 + m] = centres[u] + 0.5 * torch.randn(m, d, generator=g, device=dev)
g.manual_seed(7)
uq = torch.randint(0, k, (B,), generator=g, device=dev)
queries = (centres[uq] + 0.5 * torch.randn(B, d, generator=g, device=dev)).contiguous()
if os.environ.get("SORTQ"):  # queries grouped by the cluster they were drawn from: does the rerank gather like locality?
    queries = queries[torch.argsort(uq)].contiguous()
P = synth.random_orthogonal(d, seed=99)
idx = rabitq_amd.RaBitQ.build_device(x.data_ptr(), n, d, centres.data_ptr(), k, orthogonal=P)
del x
torch.cuda.empty_cache()
out_d = torch.empty((B, topk), device=dev)
out_i = torch.zeros((B, topk), device=dev, dtype=torch.int32)
out_n = torch.zeros((B,), device=dev, dtype=torch.int32)
rqi.set_profiling(True)
for gopt in [int(v) for v in os.environ.get("GROWTH", "").split(",") if v]:
    rqi.set_option("stage_growth", gopt)
    acc = {}
    for it in range(4):
        idx.query_batch_device(queries.data_ptr(), B, d, nprobe, topk, out_d.data_ptr(), out_i.data_ptr(), out_n.data_ptr())
        if it:
            for key, v in rqi.last_profile().items():
                acc[key] = acc.get(key, 0) + v / 3
    print(f"stage_growth={gopt}: total {acc['ms_total']:.3f} ms  scan {acc['ms_scan']:.3f} rerank {acc['ms_rerank']:.3f} replay {acc['ms_replay']:.3f} "
          f"group {acc['ms_group']:.3f} sort {acc['ms_sort']:.3f} launches {acc['scan_launches']:.0f} cand/query {acc['rerank_candidates'] / B:.1f}", flush=True)
rqi.set_option("stage_growth", 0)
modes = [int(v) for v in os.environ.get("MODES", "0,1,2,3,0").split(",") if v]
for mode in modes:
    rqi.set_option("scan_debug", mode)
    acc = {}
    for it in range(4):
        idx.query_batch_device(queries.data_ptr(), B, d, nprobe, topk, out_d.data_ptr(), out_i.data_ptr(), out_n.data_ptr())
        if it:
            for key, v in rqi.last_profile().items():
                acc[key] = acc.get(key, 0) + v / 3
    print(f"scan_debug={mode}: scan {acc['ms_scan']:.3f} ms  rerank {acc['ms_rerank']:.3f}  total {acc['ms_total']:.3f}  "
          f"rerank candidates/query {acc['rerank_candidates'] / B:.1f}", flush=True)
    print("   ", {k[3:]: round(v, 3) for k, v in acc.items() if k.startswith("ms_")}, flush=True)
rqi.set_option("scan_debug", 0)
