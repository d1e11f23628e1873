"""Developer tool: per-phase device time of one-query calls (profiling level 1), 100M x 128 by default."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rabitq_amd
from rabitq_amd import index as rqi
from tests import synth
n, d, k, nprobe, topk = int(os.environ.get("N", 100_000_000)), 128, 4096, 64, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
centres = torch.randn(k, d, generator=g, device=dev)
x = torch.empty((n, d), device=dev)
for ci, i0 in enumerate(range(0, n, 4_000_000)):
    m = min(4_000_000, n - i0); g.manual_seed(42 + ci)
    u = torch.randint(0, k, (m,), generator=g, device=dev)
    x[i0:i0 + m] = centres[u] + 0.5 * torch.randn(m, d, generator=g, device=dev)
g.manual_seed(7)
uq = torch.randint(0, k, (64,), generator=g, device=dev)
queries = (centres[uq] + 0.5 * torch.randn(64, d, generator=g, device=dev)).cpu().numpy()
idx = rabitq_amd.RaBitQ.build_device(x.data_ptr(), n, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, seed=99))
del x
for level in (0, 2, 1):
    rqi.set_profiling(level)
    for q in queries[:8]:
        idx.query(q, nprobe, topk)
    acc, t0 = {}, time.perf_counter()
    for q in queries:
        idx.query(q, nprobe, topk)
        if level:
            for key, v in rqi.last_profile().items():
                acc[key] = acc.get(key, 0) + v / len(queries)
    wall = (time.perf_counter() - t0) / len(queries) * 1e3
    print(f"profiling level {level}: wall {wall:.4f} ms/query", {k_[3:]: round(v, 4) for k_, v in acc.items() if k_.startswith("ms_")}, flush=True)
