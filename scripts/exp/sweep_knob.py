"""batch sweep under one value of an engine knob: python scripts/exp/sweep_knob.py name=value [sizes...] (100M x 128 index)"""
import sys, os, json, time
sys.path.insert(0, os.getcwd())
import torch
import rabitq_amd
from rabitq_amd import _lib, index as rqi
from tests import synth
import bench
_lib.check(_lib.lib().rq_init(0))
dev = torch.device("cuda", 0)
n, d, k, nprobe, topk, sigma = 100_000_000, 128, 4096, 64, 10, 0.5
centres = synth.device_centres(k, d, dev, 1.0)
P = synth.random_orthogonal(d, seed=99)
builder = rabitq_amd.RaBitQ.builder(n, d, centres.data_ptr(), k, orthogonal=P, max_device_base_bytes=0)
chunk = 4_000_000
chunks = [(ci, i0, min(chunk, n - i0)) for ci, i0 in enumerate(range(0, n, chunk))]
gen = lambda ci, i0, m: synth.device_mixture_chunk(centres, i0, m, sigma, ci, 42, 0, k, None)[0]
for ci, i0, m in chunks:
    builder.assign_chunk(gen(ci, i0, m).data_ptr(), i0, m)
builder.order()
for ci, i0, m in chunks:
    xc = gen(ci, i0, m).contiguous(); torch.cuda.synchronize(); builder.place_chunk(xc.data_ptr(), i0, m); del xc
idx = builder.finish()
sizes = [int(v) for v in sys.argv[2:]] or [128, 256, 512, 1024, 2048, 4096, 8192, 16384]
for spec in sys.argv[1].split(","):
    name, _, val = spec.partition("=")
    rqi.set_option(name, int(val))
    rows = bench.batch_sweep(idx, centres, None, sigma, dev, d, nprobe, topk, sizes)
    print(spec, [(r["batch"], r["ms_per_call"], round(r["queries_per_s"])) for r in rows], flush=True)
if os.environ.get("SWEEP_PROFILE"):
    rqi.set_profiling(1)
    for b in sizes:
        qs = [synth.device_queries(centres, b, sigma, dev, seed=777 + i) for i in range(3)]
        od = torch.empty((b, topk), device=dev); oi = torch.zeros((b, topk), device=dev, dtype=torch.int32); on = torch.zeros((b,), device=dev, dtype=torch.int32)
        for q in qs:
            idx.query_batch_device(q.data_ptr(), b, d, nprobe, topk, od.data_ptr(), oi.data_ptr(), on.data_ptr())
        pr = rqi.last_profile()
        print(b, {k2[3:]: round(v, 3) for k2, v in pr.items() if k2.startswith("ms_") and v > 0.004}, "launches", pr["scan_launches"], "matrix", pr["matrix_launches"], flush=True)
