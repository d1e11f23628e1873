"""`RaBitQ`: host-side mirror of the reference's index type (src/rabitq.rs:57-68, :70-333).

Same method names, argument meaning and error behaviour as the crate: where the reference panics
(`expect` / `assert!`), these raise `RabitqError`.  Everything is delegated to librabitq_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import BuildStatsT, Info, MetricsT, ProfileT, check, lib

ARR_BASE, ARR_ORTHOGONAL, ARR_CENTROIDS, ARR_OFFSETS, ARR_MAP_IDS, ARR_CODES, ARR_FACTORS = range(7)


def _addr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class RaBitQ:
    """Device-resident RaBitQ index.  Construct with `from_path`, `build`, `load_from_dir` or
    `from_arrays`; query with `query` (one vector, like the crate) or `query_batch`."""

    def __init__(self, handle):
        self._h = handle
        info = Info()
        check(lib().rq_info(self._h, C.byref(info)))
        self.dim, self.k, self.n, self.max_list_len = int(info.dim), int(info.k), int(info.n), int(info.max_list_len)
        self.n_hbm = int(info.n_hbm)      # raw vectors in HBM; the other n - n_hbm live in pinned host memory
        self.split_rows = bool(info.split_rows)   # raw vectors stored as two 16-bit planes per row (option "split_rows")

    # ---- RaBitQ::from_path (src/rabitq.rs:159) ------------------------------------------------
    @classmethod
    def from_path(cls, base_path, centroid_path, orthogonal=None, seed: int = 0) -> "RaBitQ":
        """Build from base.fvecs + centroids.fvecs.  `orthogonal` (dim x dim, P[r][c]) fixes the
        rotation the reference draws unseeded (src/utils.rs:16-20); None = seeded Gaussian-QR."""
        h = C.c_void_p()
        P = _f32(orthogonal) if orthogonal is not None else None
        check(lib().rq_build_from_path(os.fsencode(base_path), os.fsencode(centroid_path), _addr(P), seed, C.byref(h)))
        return cls(h)

    @classmethod
    def build(cls, base, centroids, orthogonal=None, seed: int = 0) -> "RaBitQ":
        """from_path on in-memory arrays (base n x d, centroids k x d)."""
        base, centroids = _f32(base), _f32(centroids)
        if base.ndim != 2 or centroids.ndim != 2 or base.shape[1] != centroids.shape[1]:
            raise _lib.RabitqError(-2, "base and centroids must be 2-D with the same dimension (rabitq.rs:165)")
        P = _f32(orthogonal) if orthogonal is not None else None
        h = C.c_void_p()
        check(lib().rq_build(_addr(base), base.shape[0], base.shape[1], _addr(centroids), centroids.shape[0], _addr(P),
                             seed, C.byref(h)))
        return cls(h)

    @classmethod
    def build_device(cls, base_ptr: int, n: int, d: int, centroids_ptr: int, k: int, orthogonal=None,
                     seed: int = 0) -> "RaBitQ":
        """Build from device-resident arrays (raw HIP device addresses, e.g. torch.Tensor.data_ptr())."""
        P = _f32(orthogonal) if orthogonal is not None else None
        h = C.c_void_p()
        check(lib().rq_build_device(C.c_void_p(base_ptr), n, d, C.c_void_p(centroids_ptr), k, _addr(P), seed, C.byref(h)))
        return cls(h)

    @classmethod
    def builder(cls, n: int, d: int, centroids_ptr: int, k: int, orthogonal=None, seed: int = 0,
                max_device_base_bytes: int = 0) -> "Builder":
        """Streamed two-pass build for inputs that are not resident (include/rabitq_hip.h: rq_builder_*)."""
        return Builder(n, d, centroids_ptr, k, orthogonal, seed, max_device_base_bytes)

    # ---- load_from_dir / dump_to_dir (src/rabitq.rs:84, :128) ---------------------------------
    @classmethod
    def load_from_dir(cls, path) -> "RaBitQ":
        h = C.c_void_p()
        check(lib().rq_load_dir(os.fsencode(path), C.byref(h)))
        return cls(h)

    def dump_to_dir(self, path) -> None:
        check(lib().rq_dump_dir(self._h, os.fsencode(path)))

    # ---- load_from_json / dump_to_json (src/rabitq.rs:72-81) -------------------------------------
    @classmethod
    def load_from_json(cls, path) -> "RaBitQ":
        h = C.c_void_p()
        check(lib().rq_load_json(os.fsencode(path), C.byref(h)))
        return cls(h)

    def dump_to_json(self, path) -> None:
        check(lib().rq_dump_json(self._h, os.fsencode(path)))

    @classmethod
    def from_arrays(cls, base, orthogonal, centroids, offsets, map_ids, codes, factors) -> "RaBitQ":
        """From the reference's in-memory arrays (what load_from_dir produces)."""
        base, orthogonal, centroids, factors = _f32(base), _f32(orthogonal), _f32(centroids), _f32(factors)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        map_ids = np.ascontiguousarray(map_ids, dtype=np.uint32)
        codes = np.ascontiguousarray(codes, dtype=np.uint64)
        h = C.c_void_p()
        check(lib().rq_from_arrays(orthogonal.shape[0], map_ids.size, offsets.size - 1, _addr(base), _addr(orthogonal),
                                   _addr(centroids), _addr(offsets), _addr(map_ids), _addr(codes), _addr(factors),
                                   C.byref(h)))
        return cls(h)

    def rotate_device(self, x_ptr: int, n: int, out_ptr: int) -> float:
        """X' = X P on device-resident rows (MFMA kernel); returns the kernel time in ms (HIP events)."""
        ms = C.c_float()
        check(lib().rq_rotate_device(self._h, C.c_void_p(x_ptr), n, C.c_void_p(out_ptr), C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().rq_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- arrays (copies to host) ----------------------------------------------------------------
    def _get(self, which, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        check(lib().rq_get_array(self._h, which, _addr(out), out.nbytes))
        return out

    @property
    def base(self):
        return self._get(ARR_BASE, (self.n, self.dim), np.float32)

    @property
    def orthogonal(self):
        return self._get(ARR_ORTHOGONAL, (self.dim, self.dim), np.float32)

    @property
    def centroids(self):
        return self._get(ARR_CENTROIDS, (self.k, self.dim), np.float32)

    @property
    def offsets(self):
        return self._get(ARR_OFFSETS, (self.k + 1,), np.uint32)

    @property
    def map_ids(self):
        return self._get(ARR_MAP_IDS, (self.n,), np.uint32)

    @property
    def codes(self):
        return self._get(ARR_CODES, (self.n, self.dim // 64), np.uint64)

    @property
    def factors(self):
        return self._get(ARR_FACTORS, (self.n, 4), np.float32)

    def device_ptr(self, which):
        p, nbytes = C.c_void_p(), C.c_uint64()
        check(lib().rq_get_device_ptr(self._h, which, C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    # ---- RaBitQ::query (src/rabitq.rs:268) ------------------------------------------------------
    def query(self, query, probe: int, topk: int, heuristic_rank: bool = False):
        """-> list of (distance, original id), at most topk, in the reference's (unspecified,
        heap-internal) order."""
        q = _f32(query).reshape(-1)
        d = np.empty(max(topk, 1), dtype=np.float32)
        ids = np.empty(max(topk, 1), dtype=np.uint32)
        n = C.c_uint32()
        check(lib().rq_query(self._h, _addr(q), q.size, probe, topk, int(heuristic_rank), _addr(d), _addr(ids),
                             C.cast(C.byref(n), C.c_void_p)))
        return [(float(d[i]), int(ids[i])) for i in range(n.value)]

    def query_batch(self, queries, probe: int, topk: int, heuristic_rank: bool = False):
        """B queries at once -> (dist B x topk f32, ids B x topk u32, counts B u32)."""
        q = _f32(queries)
        if q.ndim != 2:
            raise _lib.RabitqError(-1, "queries must be 2-D")
        B = q.shape[0]
        d = np.full((B, max(topk, 1)), np.nan, dtype=np.float32)
        ids = np.full((B, max(topk, 1)), 0xFFFFFFFF, dtype=np.uint32)
        cnt = np.zeros(B, dtype=np.uint32)
        st = lib().rq_query_batch(self._h, _addr(q), B, q.shape[1], probe, topk, int(heuristic_rank), _addr(d),
                                  _addr(ids), _addr(cnt))
        if st != _lib.RQ_ERR_EMPTY:
            check(st)
        return d, ids, cnt

    def query_batch_device(self, q_ptr: int, nq: int, length: int, probe: int, topk: int, out_dist_ptr: int,
                           out_id_ptr: int, out_n_ptr: int, heuristic_rank: bool = False):
        """Queries and outputs already in device memory (raw addresses)."""
        check(lib().rq_query_batch_device(self._h, C.c_void_p(q_ptr), nq, length, probe, topk, int(heuristic_rank),
                                          C.c_void_p(out_dist_ptr), C.c_void_p(out_id_ptr), C.c_void_p(out_n_ptr)))

    def query_batch_device_begin(self, q_ptr: int, nq: int, length: int, probe: int, topk: int, out_dist_ptr: int,
                                 out_id_ptr: int, out_n_ptr: int, heuristic_rank: bool = False):
        """Enqueue a device-resident batch and return a ticket; finish it with query_batch_device_end(ticket).
        Batches begun back to back overlap on the device (each has its own workspace and stream)."""
        t = C.c_void_p()
        check(lib().rq_query_batch_device_begin(self._h, C.c_void_p(q_ptr), nq, length, probe, topk, int(heuristic_rank),
                                                C.c_void_p(out_dist_ptr), C.c_void_p(out_id_ptr), C.c_void_p(out_n_ptr),
                                                C.byref(t)))
        return t

    @staticmethod
    def query_batch_device_end(ticket) -> None:
        check(lib().rq_query_batch_device_end(ticket))


    # ---- sharded deployments ---------------------------------------------------------------------
    def coarse_topk_device(self, q_ptr: int, nq: int, length: int, list_lo: int, list_hi: int, probe: int,
                           out_cluster_ptr: int, out_dist_ptr: int) -> None:
        """The `probe` nearest lists among [list_lo, list_hi) per query (device buffers nq x probe)."""
        check(lib().rq_coarse_topk_device(self._h, C.c_void_p(q_ptr), nq, length, list_lo, list_hi, probe,
                                          C.c_void_p(out_cluster_ptr), C.c_void_p(out_dist_ptr)))

    def query_batch_device_probed(self, q_ptr: int, nq: int, length: int, probe_cluster_ptr: int, probe_dist_ptr: int,
                                  probe: int, topk: int, out_dist_ptr: int, out_id_ptr: int, out_n_ptr: int,
                                  heuristic_rank: bool = False) -> None:
        """query_batch_device with caller-supplied probe lists (nq x probe, visiting order)."""
        check(lib().rq_query_batch_device_probed(self._h, C.c_void_p(q_ptr), nq, length, C.c_void_p(probe_cluster_ptr),
                                                 C.c_void_p(probe_dist_ptr), probe, topk, int(heuristic_rank),
                                                 C.c_void_p(out_dist_ptr), C.c_void_p(out_id_ptr), C.c_void_p(out_n_ptr)))


    def query_batch_device_seeded(self, q_ptr: int, nq: int, length: int, probe_cluster_ptr: int, probe_dist_ptr: int,
                                  probe: int, topk: int, thr_init_ptr: int, out_dist_ptr: int, out_id_ptr: int,
                                  out_n_ptr: int, heuristic_rank: bool = False) -> None:
        """query_batch_device_probed with per-query initial thresholds (nq floats on the device, f32 max = none)."""
        check(lib().rq_query_batch_device_seeded(self._h, C.c_void_p(q_ptr), nq, length, C.c_void_p(probe_cluster_ptr),
                                                 C.c_void_p(probe_dist_ptr), probe, topk, int(heuristic_rank),
                                                 C.c_void_p(thr_init_ptr), C.c_void_p(out_dist_ptr),
                                                 C.c_void_p(out_id_ptr), C.c_void_p(out_n_ptr)))

    def partition_lists(self, world: int):
        """Greedy-by-length assignment of whole lists to `world` shards -> (owner u32[k], load u64[world])."""
        owner = np.zeros(self.k, dtype=np.uint32)
        load = np.zeros(world, dtype=np.uint64)
        check(lib().rq_partition_lists(self._h, world, _addr(owner), _addr(load)))
        return owner, load

    def shard(self, owner, rank: int) -> "RaBitQ":
        """The shard of `rank`: same centroids and rotation, only the lists with owner[c] == rank (original ids kept)."""
        owner = np.ascontiguousarray(owner, dtype=np.uint32)
        assert owner.size == self.k
        h = C.c_void_p()
        check(lib().rq_shard_index(self._h, _addr(owner), rank, C.byref(h)))
        return RaBitQ(h)

    def query_batch_sharded_device(self, comm: int, world: int, id_offset: int, q_ptr: int, nq: int, length: int,
                                   probe: int, topk: int, out_dist_ptr: int, out_id_ptr: int, out_n_ptr: int,
                                   heuristic_rank: bool = False) -> None:
        """The multi-GPU step through the C ABI: local shard query, ONE ncclAllGather on `comm` (an ncclComm_t address;
        0 when world == 1), k-way merge.  Every rank gets the same global top-k."""
        check(lib().rq_query_batch_sharded_device(self._h, C.c_void_p(comm), world, id_offset, C.c_void_p(q_ptr), nq, length,
                                                  probe, topk, int(heuristic_rank), C.c_void_p(out_dist_ptr),
                                                  C.c_void_p(out_id_ptr), C.c_void_p(out_n_ptr)))


class Builder:
    """assign_chunk every row -> order() -> place_chunk every row -> finish() -> RaBitQ.  Chunks are device pointers."""

    def __init__(self, n, d, centroids_ptr, k, orthogonal=None, seed=0, max_device_base_bytes=0):
        P = _f32(orthogonal) if orthogonal is not None else None
        self._b = C.c_void_p()
        check(lib().rq_builder_create(n, d, C.c_void_p(centroids_ptr), k, _addr(P), seed, max_device_base_bytes, C.byref(self._b)))

    def assign_chunk(self, rows_ptr: int, i0: int, m: int) -> None:
        check(lib().rq_builder_assign_chunk(self._b, C.c_void_p(rows_ptr), i0, m))

    def order(self) -> None:
        check(lib().rq_builder_order(self._b))

    def place_chunk(self, rows_ptr: int, i0: int, m: int) -> None:
        check(lib().rq_builder_place_chunk(self._b, C.c_void_p(rows_ptr), i0, m))

    def stats(self) -> dict:
        st = BuildStatsT()
        check(lib().rq_builder_stats(self._b, C.byref(st)))
        return {name: getattr(st, name) for name, _ in BuildStatsT._fields_ if name != "struct_size"}

    def finish(self) -> RaBitQ:
        h = C.c_void_p()
        b, self._b = self._b, None
        check(lib().rq_builder_finish(b, C.byref(h)))
        return RaBitQ(h)

    def __del__(self):
        if getattr(self, "_b", None):
            lib().rq_builder_free(self._b)
            self._b = None


# ---- METRICS (src/metrics.rs) --------------------------------------------------------------------
def metrics() -> dict:
    m = MetricsT()
    check(lib().rq_metrics(C.byref(m)))
    return {"query": m.query, "rough": m.rough, "precise": m.precise, "miss": m.miss}


def metrics_reset() -> None:
    check(lib().rq_metrics_reset())


def metrics_str() -> str:
    """Metrics::to_str, src/metrics.rs:30-41."""
    m = metrics()
    ratio = m["rough"] / m["precise"] if m["precise"] else float("nan")
    return (f"query: {m['query']}, rough: {m['rough']}, precise: {m['precise']}, ratio: {ratio:.2f}, "
            f"cache miss: {m['miss']}")


def set_profiling(level) -> None:
    """0/False = off, 1/True = HIP events around every kernel group, 2 = around the scan launches only."""
    check(lib().rq_set_profiling(int(level)))


def set_option(name: str, value: int) -> None:
    """Engine options (include/rabitq_hip.h: rq_set_option), e.g. set_option("scan_impl", 1)."""
    check(lib().rq_set_option(name.encode(), int(value)))


def last_profile() -> dict:
    p = ProfileT()
    check(lib().rq_last_profile(C.byref(p)))
    return {name: getattr(p, name) for name, _ in ProfileT._fields_ if name != "struct_size"}


def calculate_recall(truth, res, topk: int) -> float:
    """src/utils.rs:367-379 (host-side bookkeeping of the CLI harness, crates/cli/src/main.rs:73-74)."""
    res = list(res)
    assert len(res) == topk
    t = list(truth)[:topk]
    return sum(1 for r in res if r in t) / topk
