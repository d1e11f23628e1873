"""How many rows of the pre-filtered coarse ranking take the in-kernel exact fall-back on bench-like centroids (RQ_DEBUG_COARSE=1)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rabitq_amd
from rabitq_amd import _lib, index as ix
from tests import synth
_lib.check(_lib.lib().rq_init(0))
rng = np.random.default_rng(1)
d, k, nq = 128, 4096, 8192
centres = rng.standard_normal((k, d)).astype(np.float32)
x = (centres[rng.integers(0, k, 40 * k)] + 0.5 * rng.standard_normal((40 * k, d))).astype(np.float32)
idx = rabitq_amd.RaBitQ.build(x, centres, synth.random_orthogonal(d, seed=9))
q = torch.from_numpy((centres[rng.integers(0, k, nq)] + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)).cuda()
pc = torch.zeros((nq, 64), device="cuda", dtype=torch.int32); pd = torch.zeros((nq, 64), device="cuda")
ix.set_option("coarse_impl", 3)
for _ in range(2):
    idx.coarse_topk_device(q.data_ptr(), nq, d, 0, k, 64, pc.data_ptr(), pd.data_ptr())
