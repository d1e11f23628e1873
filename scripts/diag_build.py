import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rabitq_amd
from rabitq_amd import _lib
from tests import synth
_lib.check(_lib.lib().rq_init(0))
dev = torch.device("cuda", 0)
d, k = 128, 4096
P = synth.random_orthogonal(d, seed=99)
for n in [int(a) for a in sys.argv[1:]]:
    g = torch.Generator(device=dev); g.manual_seed(1234)
    centres = torch.randn(k, d, generator=g, device=dev)
    x = torch.empty((n, d), device=dev)
    lab = torch.empty(n, device=dev, dtype=torch.int64)
    for ci, i0 in enumerate(range(0, n, 4_000_000)):
        m = min(4_000_000, n - i0); g.manual_seed(42 + ci)
        u = torch.randint(0, k, (m,), generator=g, device=dev); lab[i0:i0+m] = u
        x[i0:i0+m] = centres[u] + 0.5 * torch.randn(m, d, generator=g, device=dev)
    idx = rabitq_amd.RaBitQ.build_device(x.data_ptr(), n, d, centres.data_ptr(), k, orthogonal=P)
    off = idx.offsets.astype(np.int64); ids = idx.map_ids
    cnt = np.bincount(ids, minlength=n)
    print(f"n={n}: offsets[-1]={off[-1]} ids max={ids.max()} dup={(cnt>1).sum()} missing={(cnt==0).sum()} zeros={(ids==0).sum()}")
    true_cnt = torch.bincount(lab, minlength=k).cpu().numpy()
    print("   list sizes equal generating-centre counts:", np.array_equal(np.diff(off), true_cnt))
    # where are the bad positions?
    bad = np.nonzero(cnt[ids] > 1)[0]
    if bad.size: print("   first/last bad positions", bad[:5], bad[-5:], "of", n, " bad lists:", np.unique(np.searchsorted(off, bad, side='right')-1)[:10])
    # base rows gathered correctly?
    ptr, nbytes = idx.device_ptr(0)
    import ctypes
    sample = np.array([0, 1, n//3, n//2, n-2, n-1])
    base = torch.empty((n, d), device=dev) if False else None
    idx.close(); del x
    torch.cuda.empty_cache()
