"""Developer tool: matrix-core scan timings under ablations (rq_set_option("scan_debug", bits)); results are wrong
under bits other than 128.   N=30000000 D=768 B=10000 MODES=0,4,64 python scripts/ablate_scan2.py
  bit 2 (4): skip the query-tile loop (per-block start-up only)   bit 6 (64): exact path off   bit 0 (1): no emission"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rabitq_amd  # noqa: E402
from rabitq_amd import index as rqi  # noqa: E402
from tests import synth  # noqa: E402

n, d, k = int(os.environ.get("N", 100_000_000)), int(os.environ.get("D", 128)), int(os.environ.get("K", 4096))
nprobe, topk, B = 64, 10, int(os.environ.get("B", 32768))
dev = torch.device("cuda", 0)
centres = synth.device_centres(k, d, dev)
queries = synth.device_queries(centres, B, 0.5, dev)
chunk = max(262_144, min(4_000_000, (512 << 20) // d))
b = rabitq_amd.RaBitQ.builder(n, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, 99))
for ps in range(2):
    for ci, i0 in enumerate(range(0, n, chunk)):
        m = min(chunk, n - i0)
        xc = synth.device_mixture_chunk(centres, i0, m, 0.5, ci)[0].contiguous()
        (b.assign_chunk if ps == 0 else b.place_chunk)(xc.data_ptr(), i0, m)
        del xc
    if ps == 0:
        torch.cuda.empty_cache()
        b.order()
idx = b.finish()
torch.cuda.empty_cache()
out_d = torch.empty((B, topk), device=dev)
out_i = torch.zeros((B, topk), device=dev, dtype=torch.int32)
out_n = torch.zeros((B,), device=dev, dtype=torch.int32)
rqi.set_profiling(2)
for mode in [int(v) for v in os.environ.get("MODES", "0,4,64,0").split(",") if v]:
    rqi.set_option("scan_debug", mode)
    acc = {}
    for it in range(4):
        idx.query_batch_device(queries.data_ptr(), B, d, nprobe, topk, out_d.data_ptr(), out_i.data_ptr(), out_n.data_ptr())
        if it:
            for key, v in rqi.last_profile().items():
                acc[key] = acc.get(key, 0) + v / 3
    print(f"n={n} d={d} B={B} scan_debug={mode}: scan_matrix {acc['ms_scan_matrix']:.3f} ms (launches {acc['matrix_launches']:.0f}) scan {acc['ms_scan']:.3f} total {acc['ms_total']:.3f}", flush=True)
rqi.set_option("scan_debug", 0)
