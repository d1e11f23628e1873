import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rabitq_amd
from rabitq_amd import _lib
from tests import synth
n, k = int(sys.argv[1]), int(sys.argv[2]); nprobe = int(sys.argv[3]); sigma=float(sys.argv[4])
d, topk, B = 128, 10, 200
dev = torch.device("cuda", 0)
_lib.check(_lib.lib().rq_init(0))
g = torch.Generator(device=dev); g.manual_seed(1234)
centres = torch.randn(k, d, generator=g, device=dev)
x = torch.empty((n, d), device=dev)
chunk = 4_000_000
labels_true = torch.empty(n, device=dev, dtype=torch.int64)
for ci, i0 in enumerate(range(0, n, chunk)):
    m = min(chunk, n - i0); g.manual_seed(42 + ci)
    u = torch.randint(0, k, (m,), generator=g, device=dev); labels_true[i0:i0+m] = u
    x[i0:i0+m] = centres[u] + sigma * torch.randn(m, d, generator=g, device=dev)
g.manual_seed(7)
uq = torch.randint(0, k, (B,), generator=g, device=dev)
queries = (centres[uq] + sigma * torch.randn(B, d, generator=g, device=dev)).contiguous()
# exact GT in float64 on a per-chunk basis using direct differences for accuracy
best_d = torch.full((B, topk), float("inf"), device=dev, dtype=torch.float64); best_i = torch.full((B, topk), -1, device=dev, dtype=torch.int64)
for i0 in range(0, n, 1_000_000):
    xb = x[i0:i0+1_000_000].double()
    d2 = torch.cdist(queries.double(), xb) ** 2
    cd, ci_ = torch.topk(d2, topk, dim=1, largest=False)
    alld = torch.cat([best_d, cd], 1); alli = torch.cat([best_i, ci_ + i0], 1)
    sel = torch.topk(alld, topk, dim=1, largest=False).indices
    best_d, best_i = torch.gather(alld, 1, sel), torch.gather(alli, 1, sel)
P = synth.random_orthogonal(d, seed=99)
idx = rabitq_amd.RaBitQ.build_device(x.data_ptr(), n, d, centres.data_ptr(), k, orthogonal=P)
off = idx.offsets.astype(np.int64); print("list len min/mean/max", np.diff(off).min(), np.diff(off).mean(), np.diff(off).max())
map_ids = torch.from_numpy(idx.map_ids.astype(np.int64)).to(dev)
# label of each original id per the index
lab_idx = torch.empty(n, dtype=torch.int64, device=dev)
offs_t = torch.from_numpy(off).to(dev)
pos_label = torch.bucketize(torch.arange(n, device=dev), offs_t[1:], right=True)
lab_idx[map_ids] = pos_label
print("fraction assigned to generating centre:", float((lab_idx == labels_true).float().mean()))
dd, ii, cnt = idx.query_batch(queries.cpu().numpy(), nprobe, topk)
gt = best_i.cpu().numpy()
rec = np.mean([len(set(ii[q,:cnt[q]].tolist()) & set(gt[q].tolist()))/topk for q in range(B)])
print("recall", rec)
from rabitq_amd import ops
y, cl, cd = ops.coarse_rank(idx, queries.cpu().numpy(), nprobe)
for q in range(3):
    gl = lab_idx[best_i[q]].cpu().numpy()
    print("q", q, "own centre", int(uq[q]), "probed[0:4]", cl[q,:4], "GT labels", gl, "in probed:", np.isin(gl, cl[q]))
    print("   GT d", best_d[q].cpu().numpy().round(3)); print("   engine d", np.sort(dd[q,:cnt[q]]).round(3))
