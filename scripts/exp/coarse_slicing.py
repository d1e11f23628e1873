"""Which slicing of the multi-GPU step's coarse ranking is cheaper per rank (world 8, k = 32 768 lists, 65 536 queries)?
  by lists   : every rank ranks its k/world lists for ALL queries (then all-gather of world x nq x probe keys + merge)
  by queries : every rank ranks ALL lists for nq/world queries (then all-gather of nq x probe keys, no merge)
Both through rq_coarse_topk_device on one GPU (the collectives are not part of this measurement)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import rabitq_amd
from rabitq_amd import _lib
from tests import synth

_lib.check(_lib.lib().rq_init(0))
dev = torch.device("cuda", 0)
for k, world in ((8192, 2), (32768, 8), (16384, 4), (8192, 2), (8192, 4)):
    d, n, nq, probe = 128, 2_000_000, 65536, 64
    centres = synth.device_centres(k, d, dev)
    x = synth.device_mixture_chunk(centres, 0, n, 0.5, 0)[0].contiguous()
    idx = rabitq_amd.RaBitQ.build_device(x.data_ptr(), n, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, 5)) if hasattr(rabitq_amd.RaBitQ, "build_device") else None
    if idx is None:
        b = rabitq_amd.RaBitQ.builder(n, d, centres.data_ptr(), k, orthogonal=synth.random_orthogonal(d, 5))
        b.assign_chunk(x.data_ptr(), 0, n); b.order(); b.place_chunk(x.data_ptr(), 0, n); idx = b.finish()
    q = synth.device_queries(centres, nq, 0.5, dev)
    pc = torch.zeros((nq, probe), device=dev, dtype=torch.int32)
    pd = torch.zeros((nq, probe), device=dev, dtype=torch.float32)
    def run(nqq, lo, hi, reps=5):
        for _ in range(2):
            idx.coarse_topk_device(q.data_ptr(), nqq, d, lo, hi, probe, pc.data_ptr(), pd.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            idx.coarse_topk_device(q.data_ptr(), nqq, d, lo, hi, probe, pc.data_ptr(), pd.data_ptr())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    print(f"   second slice: {run(nq, k // world, 2 * (k // world)):.3f} ms; half the queries by lists {run(nq // 2, 0, k // world):.3f} ms")
    print(f"k={k} world={world}: by lists ({nq} queries x {k // world} lists) {run(nq, 0, k // world):.3f} ms;  "
          f"by queries ({nq // world} queries x {k} lists) {run(nq // world, 0, k):.3f} ms", flush=True)
    idx.close(); del x, q
    torch.cuda.empty_cache()
