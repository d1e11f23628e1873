import sys, os, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rabitq_amd
from rabitq_amd import _lib
from tests import synth
n, k, d, nprobe, topk, B = 100_000_000, 4096, 128, 64, 10, 10000
if len(sys.argv) > 1: n = int(sys.argv[1])
dev = torch.device("cuda", 0); _lib.check(_lib.lib().rq_init(0))
g = torch.Generator(device=dev); g.manual_seed(1234)
centres = torch.randn(k, d, generator=g, device=dev)
x = torch.empty((n, d), device=dev)
for ci, i0 in enumerate(range(0, n, 4_000_000)):
    m = min(4_000_000, n - i0); g.manual_seed(42 + ci)
    u = torch.randint(0, k, (m,), generator=g, device=dev)
    x[i0:i0+m] = centres[u] + 0.5 * torch.randn(m, d, generator=g, device=dev)
g.manual_seed(7)
uq = torch.randint(0, k, (B,), generator=g, device=dev)
q = (centres[uq] + 0.5 * torch.randn(B, d, generator=g, device=dev)).contiguous()
idx = rabitq_amd.RaBitQ.build_device(x.data_ptr(), n, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, 99))
del x; torch.cuda.empty_cache()
od = torch.empty((B, topk), device=dev); oi = torch.zeros((B, topk), device=dev, dtype=torch.int32); on = torch.zeros(B, device=dev, dtype=torch.int32)
def run(lo, hi):
    idx.query_batch_device(q[lo:hi].data_ptr(), hi - lo, d, nprobe, topk, od[lo:hi].data_ptr(), oi[lo:hi].data_ptr(), on[lo:hi].data_ptr())
for parts in (1, 2, 3, 4):
    bounds = [(B * i // parts, B * (i + 1) // parts) for i in range(parts)]
    for rep in range(2):   # warm-up incl. workspace allocation
        th = [threading.Thread(target=run, args=b) for b in bounds]; [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(5):
        th = [threading.Thread(target=run, args=b) for b in bounds]; [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{parts} concurrent sub-batches: {dt*1e3:.2f} ms per {B} queries -> {B/dt:.0f} QPS", flush=True)
