"""world_size-2 gloo test of the multi-GPU fan-out (rabitq_amd/sharding.py): each rank indexes half
of the vectors under the shared centroid set (here with the CPU oracle standing in for a GPU
shard), the per-shard top-k are all-gathered once and merged; the merge must equal the top-k of
the union, on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from rabitq_amd.sharding import merge_shard_topk, pack_topk
    from tests import synth
    n, d, k, topk, probe = 3000, 64, 12, 10, 12
    x, centres, _ = synth.mixture(n, d, k, sigma=0.8, seed=3)
    P = synth.random_orthogonal(d, seed=4)
    queries, _, _ = synth.mixture(16, d, k, sigma=0.8, seed=5)
    lo, hi = rank * n // world, (rank + 1) * n // world
    idx = oracle.OracleIndex.build(x[lo:hi], centres, P)                 # this rank's shard
    dd = np.full((len(queries), topk), np.nan, np.float32)
    ii = np.zeros((len(queries), topk), np.int64)
    cnt = np.zeros(len(queries), np.int64)
    for qi, q in enumerate(queries):
        a, b = idx.query(q, probe, topk)
        cnt[qi] = len(b)
        dd[qi, :len(b)], ii[qi, :len(b)] = a, b
    if rank == 1:
        cnt[3] = 4                                                       # a ragged shard result
    payload = pack_topk(torch.from_numpy(dd), torch.from_numpy(ii), torch.from_numpy(cnt), id_offset=lo)
    md, mi, mc = merge_shard_topk(payload, topk)
    # sharded coarse ranking: each rank owns half of the lists
    from rabitq_amd.sharding import merge_probe_lists
    klo, khi = rank * k // world, (rank + 1) * k // world
    npq = 5
    pcl = np.full((len(queries), npq), -1, np.int32)
    pdl = np.full((len(queries), npq), np.inf, np.float32)
    full_c, full_d = [], []
    for qi, q in enumerate(queries):
        y = idx.rotate_query(q)
        cl, cd = idx.coarse_rank(y, k)
        full_c.append(cl[:npq]), full_d.append(cd[:npq])
        own = [(dv, cv) for cv, dv in zip(cl, cd) if klo <= cv < khi][:npq]
        for t, (dv, cv) in enumerate(own):
            pcl[qi, t], pdl[qi, t] = cv, dv
    gc, gd = merge_probe_lists(torch.from_numpy(pcl), torch.from_numpy(pdl), npq)
    assert np.array_equal(gc.numpy().view(np.uint32), np.array(full_c, np.uint32)), "merged probe lists differ from the single-index ranking"
    assert np.array_equal(gd.numpy().view(np.uint32), np.array(full_d, np.float32).view(np.uint32))
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), d=md.numpy(), i=mi.numpy(), c=mc.numpy(), sd=dd, si=ii + lo, sc=cnt)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_merge(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{i}.npz") for i in range(world)]
    assert np.array_equal(r[0]["d"].view(np.uint32), r[1]["d"].view(np.uint32)) and np.array_equal(r[0]["i"], r[1]["i"])
    nq, topk = r[0]["d"].shape
    for q in range(nq):
        cand = []
        for s in r:
            cand += [(float(s["sd"][q, j]), int(s["si"][q, j])) for j in range(int(s["sc"][q]))]
        cand.sort()
        want = cand[:topk]
        got = [(float(r[0]["d"][q, j]), int(r[0]["i"][q, j])) for j in range(int(r[0]["c"][q]))]
        assert got == want, q
        assert all(got[j][0] <= got[j + 1][0] for j in range(len(got) - 1))
    # the merged answer is (nearly) the exact answer over the union of the shards
    sys.path.insert(0, ROOT)
    from tests import synth
    x, _, _ = synth.mixture(3000, 64, 12, sigma=0.8, seed=3)
    queries, _, _ = synth.mixture(16, 64, 12, sigma=0.8, seed=5)
    gt = synth.brute_force_topk(x, queries, topk)
    hits = sum(len(set(r[0]["i"][q, :int(r[0]["c"][q])].tolist()) & set(gt[q].tolist())) for q in range(nq) if q != 3)
    assert hits / (10 * (nq - 1)) >= 0.95


def _shard_arrays(full, owner, rank):
    """The arrays of rank's shard as rq_shard_index cuts them: only the lists with owner[c] == rank keep their members (cluster
    order and original ids kept), every centroid stays (all ranks rank over all lists)."""
    off = full.offsets.astype(np.int64)
    keep = np.concatenate([np.arange(off[c], off[c + 1]) for c in range(len(owner)) if owner[c] == rank] or [np.zeros(0, np.int64)]).astype(np.int64)
    lens = np.where(owner == rank, np.diff(off), 0)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    W = full.dim // 64
    return dict(dim=full.dim, base=full.base[keep], orthogonal=full.orthogonal, centroids=full.centroids, offsets=offsets,
                map_ids=full.map_ids[keep], codes=full.codes.reshape(-1, W)[keep].reshape(-1), factors=full.factors.reshape(-1, 4)[keep])


def _worker_whole_lists(rank, world, port, out_dir):
    """north_star's partitioning: every rank owns WHOLE IVF lists (rq_partition_lists' rule restated in numpy:
    sharding.partition_lists), knows all centroids, ranks every query over all lists and scans only what it owns; one all-gather
    of the per-shard top-k (ORIGINAL ids: no offset), k-way merge."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from rabitq_amd.sharding import merge_shard_topk, pack_topk, partition_lists
    from tests import synth
    n, d, k, topk, probe = 6000, 64, 16, 10, 6
    x, centres, _ = synth.mixture(n, d, k, sigma=0.8, seed=13)
    x = x[np.random.default_rng(1).permutation(n)[: n - 700]]          # unequal lists
    P = synth.random_orthogonal(d, seed=14)
    queries, _, _ = synth.mixture(40, d, k, sigma=0.8, seed=15)
    full = oracle.OracleIndex.build(x, centres, P)                       # (every rank builds the same index; a real rank holds only its shard)
    owner, load = partition_lists(full.offsets, world)
    assert int(load.sum()) == len(x) and abs(int(load[0]) - int(load[1])) <= int(np.diff(full.offsets.astype(np.int64)).max())
    shard = oracle.OracleIndex.view(**_shard_arrays(full, owner, rank))
    dd = np.full((len(queries), topk), np.nan, np.float32)
    ii = np.zeros((len(queries), topk), np.int64)
    cnt = np.zeros(len(queries), np.int64)
    for qi, q in enumerate(queries):
        a, b = shard.query(q, probe, topk)                               # the probe list names lists of BOTH shards: the others are empty here
        cnt[qi] = len(b)
        dd[qi, :len(b)], ii[qi, :len(b)] = a, b
    md, mi, mc = merge_shard_topk(pack_topk(torch.from_numpy(dd), torch.from_numpy(ii), torch.from_numpy(cnt), id_offset=0), topk)
    want_d = np.full((len(queries), topk), np.nan, np.float32)
    want_i = np.zeros((len(queries), topk), np.int64)
    for qi, q in enumerate(queries):
        a, b = full.query(q, probe, topk)
        o = np.lexsort((b, a))
        want_d[qi, :len(b)], want_i[qi, :len(b)] = a[o], b[o]
    np.savez(os.path.join(out_dir, f"w{rank}.npz"), d=md.numpy(), i=mi.numpy(), c=mc.numpy(), wd=want_d, wi=want_i, owner=owner)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_own_whole_lists(tmp_path):
    world = 2
    mp.spawn(_worker_whole_lists, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"w{i}.npz") for i in range(world)]
    assert np.array_equal(r[0]["owner"], r[1]["owner"]) and set(r[0]["owner"].tolist()) == {0, 1}
    assert np.array_equal(r[0]["d"].view(np.uint32), r[1]["d"].view(np.uint32)) and np.array_equal(r[0]["i"], r[1]["i"])   # every rank: the same answer
    md, mi, mc, wd, wi = r[0]["d"], r[0]["i"], r[0]["c"], r[0]["wd"], r[0]["wi"]
    same = 0
    for q in range(md.shape[0]):
        n_ = int(mc[q])
        assert n_ == md.shape[1]
        # per-shard thresholds are looser than the single index's sequential one: a shard re-ranks a superset, so the merged top-k
        # is never worse than the single index's (and differs from it only where the lower bound was violated on a true neighbour)
        assert (md[q, :n_] <= wd[q, :n_]).all(), q
        same += len(set(mi[q, :n_].tolist()) & set(wi[q, :n_].tolist()))
    assert same >= 0.99 * md.size


def test_rccl_unique_id_survives_nul_bytes():
    """The 128-byte ncclUniqueId travels from rank 0 to the others as a Python bytes object (sharding.RcclComm); ids contain
    NUL bytes, and a c_char array field read stops at the first one (round 3: every rank but 0 got a truncated id)."""
    import ctypes as C
    from rabitq_amd import sharding
    raw = bytes([3, 0, 0, 9] + [0] * 60 + list(range(64)))
    uid = sharding._UniqueId()
    C.memmove(C.addressof(uid), raw, 128)
    assert len(bytes(uid.internal)) < 128          # the trap
    assert sharding._uid_to_bytes(uid) == raw
    back = sharding._uid_from_bytes(raw)
    assert C.string_at(C.addressof(back), 128) == raw
    import pytest
    with pytest.raises(RuntimeError):
        sharding._uid_from_bytes(raw[:100])
