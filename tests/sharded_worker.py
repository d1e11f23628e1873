"""One rank of the two-rank HIP sharded step (tests/test_sharded_gpu.py starts two of these on the same GPU with
the gloo backend: RCCL refuses two ranks on one device, and only the collective differs).

Every rank builds the same index from the same seeded data, carves its shard out of it, and answers the same
batch two ways:
  A  arbitrary partition (rq_partition_lists, greedy by list length), replicated coarse ranking: the shard runs the
     ordinary pipeline (lists it does not own are empty), ONE all-gather of per-shard top-k, merge;
  B  contiguous partition, sharded coarse ranking: rq_coarse_topk_device over the rank's own lists -> all-gather ->
     merged probe lists -> rq_query_batch_device_probed -> all-gather of per-shard top-k, merge;
  C  as B, with the thresholds shared between the shards (sharding.SeededShardQuery: nearest list first, one
     all-reduce(min) of the k-th best distances, the other lists seeded with it).
  D  the WHOLE step through the C ABI (rq_query_batch_sharded_device: handshake, sliced coarse ranking + probe-key
     all-gather, nearest list, all-reduce(min), seeded rest, top-k all-gather + status words), with the host-buffer
     transport of sharding.HostCollectives standing in for RCCL; d0 = the same entry on the shards' own thresholds.
     D must equal C, and d0 must equal B, bit for bit: same partition, same arithmetic, only the plumbing differs.
Rank 0 writes the merged results to argv[1] (npz).  argv[2] = "big": ~2M vectors over 512 lists generated on the
device (per-shard thresholds differ there and the seeded step returns short shards)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case_data():
    from tests import synth
    n, d, k, nq = 20000, 128, 40, 200
    x, centres, _ = synth.mixture(n, d, k, sigma=0.9, seed=71, centre_scale=0.6)     # overlapping clusters
    rng = np.random.default_rng(72)
    sizes = rng.zipf(1.5, size=k).clip(1, 60)                                        # unbalanced lists
    x[: int(n * 0.3)] = (centres[rng.choice(k, int(n * 0.3), p=sizes / sizes.sum())] +
                         0.9 * rng.standard_normal((int(n * 0.3), d))).astype(np.float32)
    queries, _, _ = synth.mixture(nq, d, k, sigma=0.9, seed=73, centre_scale=0.6)
    return x, centres, synth.random_orthogonal(d, seed=74), queries, 12, 10


def big_case_data(dev):
    """2M x 128 over 512 overlapping, Zipf-sized lists, 2000 queries (device tensors)."""
    import torch
    from tests import synth
    n, d, k, nq, sigma = 2_000_000, 128, 512, 2000, 0.9
    centres = synth.device_centres(k, d, dev, 0.6, seed=81)
    wz = 1.0 / torch.arange(1, k + 1, device=dev, dtype=torch.float64) ** 0.7
    w = (wz / wz.sum()).float()[torch.randperm(k, device=dev, generator=torch.Generator(device=dev).manual_seed(82))]
    x = synth.device_mixture_chunk(centres, 0, n, sigma, 0, 83, 0, k, w)[0].contiguous()
    q = synth.device_queries(centres, nq, sigma, dev, seed=84, weights=w)
    return x, centres, synth.random_orthogonal(d, seed=85), q, 32, 10


def main():
    out_path = sys.argv[1]
    big = len(sys.argv) > 2 and sys.argv[2] == "big"
    import torch
    import torch.distributed as dist
    import rabitq_amd
    from rabitq_amd import _lib, sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib.check(_lib.lib().rq_init(0))
    dev = torch.device("cuda", 0)
    if big:
        xd, cd, P, q, probe, topk = big_case_data(dev)
        n, d = xd.shape
        k, nq = cd.shape[0], q.shape[0]
        full = rabitq_amd.RaBitQ.build_device(xd.data_ptr(), n, d, cd.data_ptr(), k, orthogonal=P)
        del xd
        torch.cuda.empty_cache()
    else:
        x, centres, P, queries, probe, topk = case_data()
        n, d = x.shape
        k, nq = centres.shape[0], queries.shape[0]
        full = rabitq_amd.RaBitQ.build(x, centres, P)
        q = torch.from_numpy(queries).to(dev)
    od = torch.empty((nq, topk), device=dev)
    oi = torch.zeros((nq, topk), device=dev, dtype=torch.int32)
    on = torch.zeros(nq, device=dev, dtype=torch.int32)
    res = {}

    # ---- A: greedy partition, replicated coarse ranking ------------------------------------------------------
    owner, load = full.partition_lists(world)
    sh = full.shard(owner, rank)
    assert sh.n == int(load[rank]) and sh.k == k
    sh.query_batch_device(q.data_ptr(), nq, d, probe, topk, od.data_ptr(), oi.data_ptr(), on.data_ptr())
    pay = sharding.pack_topk(od, oi.to(torch.int64) & 0xFFFFFFFF, on, 0).cpu()      # shard map_ids are global already
    md, mi, mn = sharding.merge_shard_topk(pay, topk, id_bound=n)
    res.update(a_dist=md.numpy(), a_ids=mi.numpy(), a_cnt=mn.numpy(), owner=owner, load=load)
    sh.close()

    # ---- B: contiguous halves, sharded coarse ranking + caller-supplied probe lists ---------------------------
    bounds = [(r * k) // world for r in range(world + 1)]
    owner_b = np.zeros(k, np.uint32)
    for r in range(world):
        owner_b[bounds[r]:bounds[r + 1]] = r
    shb = full.shard(owner_b, rank)
    pc = torch.zeros((nq, probe), device=dev, dtype=torch.int32)
    pdd = torch.zeros((nq, probe), device=dev, dtype=torch.float32)
    shb.coarse_topk_device(q.data_ptr(), nq, d, bounds[rank], bounds[rank + 1], probe, pc.data_ptr(), pdd.data_ptr())
    mc, mdist = sharding.merge_probe_lists(pc.cpu(), pdd.cpu(), probe)               # one all-gather (gloo)
    mc_d, md_d = mc.to(dev), mdist.to(dev)
    torch.cuda.synchronize()
    shb.query_batch_device_probed(q.data_ptr(), nq, d, mc_d.data_ptr(), md_d.data_ptr(), probe, topk, od.data_ptr(),
                                  oi.data_ptr(), on.data_ptr())
    b_rerank = rabitq_amd.index.last_profile()["rerank_candidates"]
    pay = sharding.pack_topk(od, oi.to(torch.int64) & 0xFFFFFFFF, on, 0).cpu()
    bd, bi, bn = sharding.merge_shard_topk(pay, topk, id_bound=n)
    res.update(b_dist=bd.numpy(), b_ids=bi.numpy(), b_cnt=bn.numpy(), b_probe=mc.numpy().view(np.uint32),
               b_probe_dist=mdist.numpy(), b_local_rerank=np.array([b_rerank]))
    # ---- C: the same partition and probe lists, thresholds shared between the shards (two engine calls) --------
    sq = sharding.SeededShardQuery(nq, topk, dev)
    sq.run(shb, q.data_ptr(), d, mc_d, md_d, cpu_collectives=True)
    cd, ci, cn = sharding.merge_shard_topk(sq.payload(0).cpu(), topk, id_bound=n)
    res.update(c_dist=cd.numpy(), c_ids=ci.numpy(), c_cnt=cn.numpy(), c_thr=sq.thr.cpu().numpy(),
               c_local_rerank=np.array([sq.profile_a["rerank_candidates"] + rabitq_amd.index.last_profile()["rerank_candidates"]]))
    # ---- D: the whole step through the C ABI, host-buffer collectives instead of RCCL -------------------------------
    hc = sharding.HostCollectives()
    hc.install()
    for tag, shared in (("d", 1), ("d0", 0)):
        rabitq_amd.index.set_option("shared_thresholds", shared)
        hc.calls.clear()
        od.fill_(-1.0), oi.zero_(), on.zero_()
        shb.query_batch_sharded_device(1, world, 0, q.data_ptr(), nq, d, probe, topk, od.data_ptr(), oi.data_ptr(), on.data_ptr())
        res.update({f"{tag}_dist": od.cpu().numpy(), f"{tag}_ids": oi.cpu().numpy().view(np.uint32).astype(np.int64),
                    f"{tag}_cnt": on.cpu().numpy().astype(np.int64),
                    f"{tag}_local_rerank": np.array([rabitq_amd.index.last_profile()["rerank_candidates"]])})
        per_rank = -(-nq // world) * probe   # the coarse ranking is sliced by queries: a rank contributes its queries' probe lists
        want = ([("all_reduce", 4), ("all_gather", per_rank), ("all_reduce", nq), ("all_gather", nq * 2 * topk + 1)] if shared else
                [("all_reduce", 4), ("all_gather", per_rank), ("all_gather", nq * topk + 1)])
        assert hc.calls == want, (hc.calls, want)
    rabitq_amd.index.set_option("shared_thresholds", 1)
    # a rank called with other parameters: the handshake makes EVERY rank fail, nobody blocks in a collective
    try:
        shb.query_batch_sharded_device(1, world, 0, q.data_ptr(), nq, d, probe, topk + rank, od.data_ptr(), oi.data_ptr(), on.data_ptr())
        raise SystemExit("parameter mismatch between the ranks was not detected")
    except rabitq_amd.RabitqError as e:
        assert e.status == -1 and "different" in str(e), str(e)
    # ... and a rank that fails validation (null output on rank 1 only): the others return an error too
    try:
        shb.query_batch_sharded_device(1, world, 0, q.data_ptr(), nq, d, probe, topk, od.data_ptr() if rank == 0 else 0,
                                       oi.data_ptr(), on.data_ptr())
        raise SystemExit("a failed rank was not reported to its peers")
    except rabitq_amd.RabitqError as e:
        assert e.status == -1, str(e)
    sharding.HostCollectives.uninstall()
    shb.close()
    full.close()
    dist.barrier()
    if rank == 0:
        np.savez(out_path, **res)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
