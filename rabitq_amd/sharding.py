"""Multi-GPU query fan-out: IVF lists (or, equivalently, the vectors under a shared centroid set)
are sharded one shard per GPU / process; every rank answers the same query batch against its shard
and the per-shard top-k are exchanged with ONE all-gather per batch and merged.

The payload is nq * topk * 8 bytes per rank (80 B/query at topk = 10), i.e. latency-bound on xGMI,
so a single fused all-gather per batch is the right collective -- not a ring all-reduce
(SURVEY.md section 8e).  torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU) is plumbing.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist


def _engine_merge(gathered: torch.Tensor, world: int, nq: int, width: int, m_out: int) -> torch.Tensor:
    """Per query the m_out smallest u64 keys of gathered[world][nq][width] (int64 tensor holding the bit
    patterns, on the GPU), ascending: the engine's merge kernel (rq_merge_smallest_u64_device), launched on
    the default stream like torch's own kernels."""
    from ._lib import check, lib
    out = torch.empty((nq, m_out), dtype=torch.int64, device=gathered.device)
    check(lib().rq_merge_smallest_u64_device(C.c_void_p(gathered.data_ptr()), world, nq, width, m_out,
                                             C.c_void_p(out.data_ptr())))
    return out


def _use_engine(t: torch.Tensor, world: int, width: int) -> bool:
    return t.is_cuda and world * width <= 16384 and torch.cuda.current_stream(t.device).cuda_stream == 0


def partition_lists(offsets, world: int):
    """The partitioning rule of rq_partition_lists restated on the host (numpy only): whole lists to shards, greedy by list
    length -- longest first (ties: lower list id first), each to the least-loaded shard (ties: lower shard first).  Deterministic,
    so every rank computes the same owner table on its own.  -> (owner u32[k], load u64[world])"""
    import heapq
    import numpy as np
    off = np.asarray(offsets, dtype=np.int64)
    lens = np.diff(off)
    k = lens.size
    owner = np.zeros(k, dtype=np.uint32)
    heap = [(0, r) for r in range(world)]
    for c in sorted(range(k), key=lambda c: (-int(lens[c]), c)):
        load, r = heapq.heappop(heap)
        owner[c] = r
        heapq.heappush(heap, (load + int(lens[c]), r))
    load = np.zeros(world, dtype=np.uint64)
    for c in range(k):
        load[owner[c]] += np.uint64(lens[c])
    return owner, load


def pack_topk(dist_t: torch.Tensor, ids_t: torch.Tensor, counts: torch.Tensor, id_offset: int) -> torch.Tensor:
    """(nq, topk) f32 distances + u32/i64 local ids + valid counts -> (nq, topk, 2) i64 payload of
    (monotone distance key, global id); invalid entries get the maximum key."""
    nq, topk = dist_t.shape
    bits = dist_t.contiguous().view(torch.int32).to(torch.int64)
    key = torch.where(bits < 0, bits ^ 0x7FFFFFFF, bits)          # Ord32 (src/ord32.rs:12-17)
    valid = torch.arange(topk, device=dist_t.device)[None, :] < counts.to(torch.int64)[:, None]
    key = torch.where(valid, key, torch.full_like(key, 2**31))
    gid = ids_t.to(torch.int64) + id_offset
    gid = torch.where(valid, gid, torch.full_like(gid, -1))
    return torch.stack([key, gid], dim=-1)


def merge_shard_topk(payload: torch.Tensor, topk: int, group=None, id_bound: int | None = None):
    """All-gather every rank's (nq, topk, 2) payload and keep the topk smallest per query.
    Returns (dist f32 (nq, topk), ids i64 (nq, topk), counts i64 (nq,)), identical on every rank.
    id_bound: if every global id is below it and it fits 32 bits, a (key, id) pair travels as ONE u64 (half the
    all-gather payload) and the merge runs in the engine's kernel on the GPU; otherwise in torch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq = payload.shape[0]
    if id_bound is not None and id_bound < 2**32 - 1 and _use_engine(payload, world, payload.shape[1]):
        key, gid = payload[..., 0], payload[..., 1]
        valid = key < 2**31
        k32 = torch.where(valid, key + 2**31, torch.full_like(key, 0xFFFFFFFF))      # monotone u32 image, invalid last
        g32 = torch.where(valid, gid, torch.full_like(gid, 0xFFFFFFFF)) & 0xFFFFFFFF
        packed = ((k32 << 32) | g32).contiguous()                                     # bit pattern of the u64 key
        if world > 1:
            gathered = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
            dist.all_gather_into_tensor(gathered.reshape(-1), packed.reshape(-1), group=group)
        else:
            gathered = packed[None]
        out = _engine_merge(gathered, world, nq, packed.shape[1], topk)
        k32o, gido = (out >> 32) & 0xFFFFFFFF, out & 0xFFFFFFFF
        ok = ~((k32o == 0xFFFFFFFF) & (gido == 0xFFFFFFFF))
        counts = ok.sum(dim=1)
        key_o = (k32o - 2**31).clamp(min=-2**31, max=2**31 - 1).to(torch.int32)
        bits = torch.where(key_o < 0, key_o ^ 0x7FFFFFFF, key_o)
        return bits.view(torch.float32), torch.where(ok, gido, torch.full_like(gido, -1)), counts
    if world > 1:
        flat = payload.contiguous().reshape(-1)
        gathered = torch.empty(world * flat.numel(), dtype=payload.dtype, device=payload.device)
        dist.all_gather_into_tensor(gathered, flat, group=group)   # ONE collective per query batch
        gathered = gathered.reshape((world,) + tuple(payload.shape))
    else:
        gathered = payload[None]
    allp = gathered.permute(1, 0, 2, 3).reshape(nq, -1, 2)       # (nq, world*topk, 2)
    # order by (key, global id): deterministic on every rank
    by_id = torch.argsort(allp[..., 1], dim=1, stable=True)      # two stable passes = lexicographic
    k1 = torch.gather(allp[..., 0], 1, by_id)
    by_key = torch.argsort(k1, dim=1, stable=True)[:, :topk]
    order = torch.gather(by_id, 1, by_key)
    key = torch.gather(allp[..., 0], 1, order)
    gid = torch.gather(allp[..., 1], 1, order)
    counts = (key < 2**31).sum(dim=1)
    k32 = key.clamp(max=2**31 - 1).to(torch.int32)
    bits = torch.where(k32 < 0, k32 ^ 0x7FFFFFFF, k32)
    return bits.view(torch.float32), gid, counts


def merge_probe_lists(cluster: torch.Tensor, dist_t: torch.Tensor, nprobe: int, group=None):
    """Sharded coarse ranking (src/rabitq.rs:283-297 over lists owned by different ranks): every rank
    passes its (nq, nprobe) nearest OWN lists (global ids as int32/uint32 bits, f32 distances, padded
    with id 0xFFFFFFFF / +inf); one all-gather, then the nprobe nearest overall per query, ascending
    by (distance, list id) -- the same order a single index would visit them in.
    Returns (cluster int32 (nq, nprobe) with the u32 bit patterns, dist f32 (nq, nprobe)).
    The result is produced on torch's current stream: synchronise it before handing the tensors to the
    engine (`rq_query_batch_device_probed` runs on its own streams)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    bits = dist_t.contiguous().view(torch.int32).to(torch.int64)          # distances are >= 0: bits order like values
    ids = cluster.to(torch.int64) & 0xFFFFFFFF
    key = (bits << 32) | ids
    nq, width = key.shape
    if world > 1:
        flat = key.contiguous().reshape(-1)
        gathered = torch.empty(world * flat.numel(), dtype=key.dtype, device=key.device)
        dist.all_gather_into_tensor(gathered, flat, group=group)
        if _use_engine(key, world, width):
            key = _engine_merge(gathered.reshape(world, nq, width), world, nq, width, nprobe)
        else:
            key = gathered.reshape(world, nq, width).permute(1, 0, 2).reshape(nq, -1)
            key = torch.sort(key, dim=1).values[:, :nprobe]
    elif _use_engine(key, 1, width):
        key = _engine_merge(key.contiguous()[None], 1, nq, width, nprobe)
    else:
        key = torch.sort(key, dim=1).values[:, :nprobe]
    out_ids = (key & 0xFFFFFFFF).to(torch.int32)      # wraps to the u32 bit pattern
    out_dist = (key >> 32).to(torch.int32).view(torch.float32)
    return out_ids.contiguous(), out_dist.contiguous()


class SeededShardQuery:
    """The per-shard part of a multi-GPU query step with SHARED thresholds ("per-shard thresholds are
    looser"): a shard that does not hold a query's neighbourhood never fills its ranker with near candidates, so on its
    own it would re-rank (and, in the matrix-core scan, re-evaluate exactly) most of what it scans.  Two engine calls:
      A  the query's NEAREST list alone (only its owner finds candidates there): the usual staged pass;
      -  threshold = the k-th best exact distance of A where A is full (an actual k-th best of a subset, hence an upper
         bound of the final k-th distance), f32 max elsewhere; ONE all-reduce(min) of nq floats gives it to every shard;
      B  the other probed lists, seeded with that threshold (rq_query_batch_device_seeded): one stage, no learning.
    The caller merges A and B of all shards (payload(): (nq, 2 topk, 2) for merge_shard_topk).  The work is the same
    candidates as one probed call; what changes is that B prunes with the best threshold any shard knows."""

    def __init__(self, nq: int, topk: int, device):
        f32, i32 = torch.float32, torch.int32
        self.nq, self.topk = nq, topk
        self.a = (torch.empty((nq, topk), device=device, dtype=f32), torch.zeros((nq, topk), device=device, dtype=i32),
                  torch.zeros((nq,), device=device, dtype=i32))
        self.b = (torch.empty((nq, topk), device=device, dtype=f32), torch.zeros((nq, topk), device=device, dtype=i32),
                  torch.zeros((nq,), device=device, dtype=i32))
        self.profile_a = None

    def run(self, idx, q_ptr: int, length: int, pc: torch.Tensor, pdist: torch.Tensor, group=None,
            cpu_collectives: bool = False):
        """pc / pdist: the merged probe lists (nq, nprobe) on the device, visiting order.  Runs A, the all-reduce and B;
        results stay in self.a / self.b (dist, local id, count)."""
        from . import index as _ix
        nq, topk = self.nq, self.topk
        nprobe = pc.shape[1]
        ad, ai, an = self.a
        bd, bi, bn = self.b
        pc_a, pd_a = pc[:, :1].contiguous(), pdist[:, :1].contiguous()
        torch.cuda.current_stream().synchronize()   # the engine runs on its own streams
        idx.query_batch_device_probed(q_ptr, nq, length, pc_a.data_ptr(), pd_a.data_ptr(), 1, topk, ad.data_ptr(),
                                      ai.data_ptr(), an.data_ptr())
        self.profile_a = _ix.last_profile()
        fmax = torch.finfo(torch.float32).max
        valid = torch.arange(topk, device=ad.device)[None, :] < an.to(torch.int64)[:, None]
        kth = torch.where(valid, ad, torch.full_like(ad, -fmax)).max(dim=1).values
        thr = torch.where(an == topk, kth, torch.full_like(kth, fmax)).contiguous()
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            if cpu_collectives:
                t = thr.cpu()
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
                thr = t.to(ad.device)
            else:
                dist.all_reduce(thr, op=dist.ReduceOp.MIN, group=group)
        self.thr = thr
        if nprobe > 1:
            pc_b, pd_b = pc[:, 1:].contiguous(), pdist[:, 1:].contiguous()
            torch.cuda.current_stream().synchronize()
            idx.query_batch_device_seeded(q_ptr, nq, length, pc_b.data_ptr(), pd_b.data_ptr(), nprobe - 1, topk,
                                          thr.data_ptr(), bd.data_ptr(), bi.data_ptr(), bn.data_ptr())
        else:
            bn.zero_()

    def payload(self, id_offset: int) -> torch.Tensor:
        """(nq, 2 topk, 2) merge payload of A and B (merge_shard_topk keeps the topk smallest over all shards)."""
        pa = pack_topk(self.a[0], self.a[1].to(torch.int64) & 0xFFFFFFFF, self.a[2], id_offset)
        pb = pack_topk(self.b[0], self.b[1].to(torch.int64) & 0xFFFFFFFF, self.b[2], id_offset)
        return torch.cat([pa, pb], dim=1)


# ---- the C-ABI multi-GPU step (rq_query_batch_sharded_device) from Python: communicators and transports --------------
class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]   # ncclUniqueId (rccl.h)


def _uid_to_bytes(uid: "_UniqueId") -> bytes:
    """All 128 bytes of the id (reading the c_char array FIELD would stop at its first NUL byte)."""
    return C.string_at(C.addressof(uid), C.sizeof(uid))


def _uid_from_bytes(raw) -> "_UniqueId":
    if not isinstance(raw, (bytes, bytearray)) or len(raw) != C.sizeof(_UniqueId):
        raise RuntimeError("the RCCL unique id did not arrive intact")
    uid = _UniqueId()
    C.memmove(C.addressof(uid), bytes(raw), C.sizeof(uid))
    return uid


def _rccl_lib():
    """The RCCL this process already carries (torch's bundled copy), so that the communicator handle and the
    collectives the engine resolves with dlsym come from the same library (RABITQ_RCCL_LIB overrides)."""
    import os
    path = os.environ.get("RABITQ_RCCL_LIB")
    cands = [path] if path else []
    cands += [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so.1", "librccl.so"]
    last = None
    for c in cands:
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:       # noqa: PERF203
            last = e
    raise OSError(f"RCCL not found: {last}")


class RcclComm:
    """An ncclComm_t of torch.distributed's world, created with ncclGetUniqueId (rank 0) / ncclCommInitRank; the 128-byte
    id travels through the already initialised process group (any backend).  `.handle` is what
    rq_query_batch_sharded_device takes.  One HIP device per rank (RCCL refuses two ranks on one device)."""

    def __init__(self, rank: int, world: int, group=None):
        self.lib = _rccl_lib()
        uid = _UniqueId()
        if rank == 0:
            rc = self.lib.ncclGetUniqueId(C.byref(uid))
            if rc != 0:
                raise RuntimeError(f"ncclGetUniqueId failed ({rc})")
        box = [_uid_to_bytes(uid)] if rank == 0 else [None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        uid = _uid_from_bytes(box[0])
        self.comm = C.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        rc = self.lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed ({rc})")
        self.handle = self.comm.value

    def close(self):
        if getattr(self, "comm", None) and self.comm.value:
            self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


class HostCollectives:
    """rq_set_collectives with torch.distributed on HOST buffers as the transport (gloo): the device buffers are copied
    to the host, exchanged, and copied back.  For rehearsals and tests on one GPU -- several rank processes share the
    device, which RCCL refuses -- where every kernel and the whole step logic are the engine's and only the transport
    differs.  Keep the object alive while it is installed."""
    _SIZES = {2: 4, 5: 8, 7: 4}
    _DTYPES = {2: torch.int32, 5: torch.int64, 7: torch.float32}

    class _Table(C.Structure):
        _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("all_gather", C.c_void_p),
                    ("all_reduce", C.c_void_p), ("comm_user_rank", C.c_void_p)]

    def __init__(self, group=None):
        import os
        self.group = group
        self.hip = None
        for name in (None, os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), "libamdhip64.so"):
            try:
                lib_ = C.CDLL(name)
                lib_.hipMemcpy, lib_.hipStreamSynchronize    # noqa: B018
                self.hip = lib_
                break
            except (OSError, AttributeError):
                continue
        if self.hip is None:
            raise OSError("libamdhip64 not found")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
        UR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int))
        self._ag, self._ar, self._ur = AG(self._all_gather), AR(self._all_reduce), UR(self._user_rank)
        self.calls = []     # (name, count) of every collective issued, for tests
        self.table = self._Table(C.sizeof(self._Table), 0, C.cast(self._ag, C.c_void_p), C.cast(self._ar, C.c_void_p),
                                 C.cast(self._ur, C.c_void_p))

    def install(self):
        from ._lib import check, lib
        check(lib().rq_set_collectives(C.byref(self.table)))

    @staticmethod
    def uninstall():
        from ._lib import check, lib
        check(lib().rq_set_collectives(None))

    def _to_host(self, ptr, count, dtype, stream):
        t = torch.empty(count, dtype=self._DTYPES[dtype])
        if self.hip.hipStreamSynchronize(stream) != 0 or self.hip.hipMemcpy(t.data_ptr(), ptr, count * self._SIZES[dtype], 2) != 0:
            raise RuntimeError("hip copy to host failed")
        return t

    def _all_gather(self, send, recv, count, dtype, comm, stream):
        try:
            self.calls.append(("all_gather", int(count)))
            t = self._to_host(send, count, dtype, stream)
            world = dist.get_world_size(self.group)
            out = torch.empty(count * world, dtype=t.dtype)
            dist.all_gather_into_tensor(out, t, group=self.group)
            return 0 if self.hip.hipMemcpy(recv, out.data_ptr(), out.numel() * self._SIZES[dtype], 1) == 0 else 1
        except Exception:       # a Python exception must not unwind through the C caller
            import traceback
            traceback.print_exc()
            return 1

    def _all_reduce(self, send, recv, count, dtype, op, comm, stream):
        try:
            self.calls.append(("all_reduce", int(count)))
            t = self._to_host(send, count, dtype, stream)
            dist.all_reduce(t, op={2: dist.ReduceOp.MAX, 3: dist.ReduceOp.MIN}[op], group=self.group)
            return 0 if self.hip.hipMemcpy(recv, t.data_ptr(), count * self._SIZES[dtype], 1) == 0 else 1
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _user_rank(self, comm, out_rank):
        try:   # an exception inside a ctypes callback is swallowed (the call would return 0 with out_rank unset: rank 0's slice)
            out_rank[0] = dist.get_rank(self.group)
            return 0
        except Exception:  # noqa: BLE001
            return 1


class EmulatedPeers(HostCollectives):
    """One-GPU rehearsal of ONE rank of a W-GPU step (bench.py --emulate-world W): the collectives of a one-rank world,
    except that the all-reduce(min) of the shared thresholds fills in what the absent peers would contribute -- a query
    whose nearest list lives on another rank gets no seed from this rank's pass A (f32::MAX); its owner would send the k-th
    best distance it found there.  On a homogeneous mixture those distances are alike for all queries, so the MEDIAN of
    the seeds this rank did find stands in for them.  Results of such a run are not parity material; the per-kernel times
    are what a real rank's step costs."""

    def _all_gather(self, send, recv, count, dtype, comm, stream):
        try:
            self.calls.append(("all_gather", int(count)))
            t = self._to_host(send, count, dtype, stream)
            return 0 if self.hip.hipMemcpy(recv, t.data_ptr(), count * self._SIZES[dtype], 1) == 0 else 1
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _all_reduce(self, send, recv, count, dtype, op, comm, stream):
        try:
            self.calls.append(("all_reduce", int(count)))
            t = self._to_host(send, count, dtype, stream)
            if dtype == 7 and op == 3:      # f32 min: the shared thresholds
                have = t < 3.0e38
                if bool(have.any()):
                    t[~have] = t[have].median()
            return 0 if self.hip.hipMemcpy(recv, t.data_ptr(), count * self._SIZES[dtype], 1) == 0 else 1
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def _user_rank(self, comm, out_rank):
        out_rank[0] = 0
        return 0
