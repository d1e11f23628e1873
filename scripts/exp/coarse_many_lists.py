"""Run ON THE GPU BOX: time of the coarse ranking (rq_coarse_topk_device: 65 536 queries against ALL lists) for the list counts
of 1 / 2 / 4 / 8-GPU deployments (4096 lists per GPU), pre-filtered (coarse_impl 0: auto) vs the exact-order kernels + selection
(coarse_impl 2), and whether the two agree bit for bit."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rabitq_amd as rq
from rabitq_amd import index as ix
from tests import synth
dev = torch.device("cuda", 0)
d, nq, probe = int(os.environ.get("D", 128)), int(os.environ.get("NQ", 65536)), 64
out = []
for k in [int(v) for v in os.environ.get("KLIST", "2048,4096,8192,16384,32768").split(",")]:
    g = torch.Generator(device=dev); g.manual_seed(k)
    centres = torch.randn(k, d, generator=g, device=dev)
    x = (centres[torch.randint(0, k, (4 * k,), generator=g, device=dev)] + 0.5 * torch.randn(4 * k, d, generator=g, device=dev)).contiguous()
    q = (centres[torch.randint(0, k, (nq,), generator=g, device=dev)] + 0.5 * torch.randn(nq, d, generator=g, device=dev)).contiguous()
    idx = rq.RaBitQ.build_device(x.data_ptr(), 4 * k, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, seed=1))
    res = {}
    for impl in (0, 3, 2):
        ix.set_option("coarse_impl", impl)
        pc = torch.zeros((nq, probe), device=dev, dtype=torch.int32)
        pd = torch.zeros((nq, probe), device=dev, dtype=torch.float32)
        for _ in range(2):
            idx.coarse_topk_device(q.data_ptr(), nq, d, 0, k, probe, pc.data_ptr(), pd.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            idx.coarse_topk_device(q.data_ptr(), nq, d, 0, k, probe, pc.data_ptr(), pd.data_ptr())
        torch.cuda.synchronize()
        res[impl] = ((time.perf_counter() - t0) / 3 * 1e3, pc.cpu().numpy().copy(), pd.cpu().numpy().view(np.uint32).copy())
    ix.set_option("coarse_impl", 0)
    same = bool(np.array_equal(res[0][1], res[2][1]) and np.array_equal(res[0][2], res[2][2]))
    same = same and bool(np.array_equal(res[3][1], res[2][1]) and np.array_equal(res[3][2], res[2][2]))
    row = {"lists": k, "prefiltered_tile_minima_ms": round(res[0][0], 3), "prefiltered_row_in_registers_ms": round(res[3][0], 3), "exact_order_ms": round(res[2][0], 3), "identical": same}
    print(json.dumps(row), flush=True)
    out.append(row)
    idx.close()
    del x, q, centres
