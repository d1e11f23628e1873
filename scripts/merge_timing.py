"""Developer tool: GPU time of the torch-side merges of the multi-GPU step at a simulated world size."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rabitq_amd import sharding
dev = torch.device("cuda", 0)
B, nprobe, topk = int(os.environ.get('B', 32768)), 64, 10
for world in (1, 2, 4, 8):
    cl = torch.randint(0, 4096 * world, (B, nprobe * world), device=dev, dtype=torch.int32)
    dd = torch.rand((B, nprobe * world), device=dev)
    d = torch.rand((B, topk * world), device=dev); i = torch.randint(0, 10**9, (B, topk * world), device=dev)
    n = torch.full((B,), topk * world, device=dev)
    for _ in range(3):
        sharding.merge_probe_lists(cl, dd, nprobe); pay = sharding.pack_topk(d, i, n, 0); sharding.merge_shard_topk(pay, topk, id_bound=2**31)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): sharding.merge_probe_lists(cl, dd, nprobe)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(10):
        pay = sharding.pack_topk(d, i, n, 0); sharding.merge_shard_topk(pay, topk, id_bound=2**31)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"world {world}: probe-list merge {(t1 - t0) * 100:.3f} ms, top-k pack+merge {(t2 - t1) * 100:.3f} ms (gathered widths {nprobe * world}, {topk * world})", flush=True)
