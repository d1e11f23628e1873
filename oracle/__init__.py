"""ctypes binding of the CPU oracle (oracle/rabitq_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under rabitq_amd/ may import this package.

Parity unpinned by the reference (it ships no tests/golden vectors and cannot be built here);
see rabitq_oracle.h for how the restatement is pinned instead.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librabitq_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "rabitq_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "rabitq_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


class Metrics(C.Structure):
    _fields_ = [("rough", C.c_uint64), ("precise", C.c_uint64), ("query", C.c_uint64), ("miss", C.c_uint64)]


class _Index(C.Structure):
    _fields_ = [("dim", C.c_uint32), ("n", C.c_uint64), ("k", C.c_uint32), ("base", _f32p),
                ("orthogonal", _f32p), ("orthogonal_t", _f32p), ("centroids", _f32p), ("offsets", _u32p),
                ("map_ids", _u32p), ("x_binary_vec", _u64p), ("factors", _f32p)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        sz = C.c_size_t
        L.rqo_l2_squared_distance.restype = C.c_float
        L.rqo_l2_squared_distance.argtypes = [_f32p, _f32p, sz]
        L.rqo_vector_dot_product.restype = C.c_float
        L.rqo_vector_dot_product.argtypes = [_f32p, _f32p, sz]
        L.rqo_min_max_residual.restype = None
        L.rqo_min_max_residual.argtypes = [_f32p, _f32p, _f32p, sz, _f32p, _f32p]
        L.rqo_scalar_quantize.restype = C.c_uint32
        L.rqo_scalar_quantize.argtypes = [_u8p, _f32p, sz, C.c_float, C.c_float]
        L.rqo_vector_binarize_query.restype = None
        L.rqo_vector_binarize_query.argtypes = [_u8p, sz, _u64p]
        L.rqo_binary_dot_product.restype = C.c_uint32
        L.rqo_binary_dot_product.argtypes = [_u64p, _u64p, sz]
        L.rqo_asymmetric_binary_dot_product.restype = C.c_uint32
        L.rqo_asymmetric_binary_dot_product.argtypes = [_u64p, _u64p, sz]
        L.rqo_vector_binarize_u64.restype = None
        L.rqo_vector_binarize_u64.argtypes = [_f32p, sz, _u64p]
        L.rqo_project.restype = None
        L.rqo_project.argtypes = [_f32p, _f32p, sz, _f32p]
        L.rqo_kmeans_nearest_cluster.restype = None
        L.rqo_kmeans_nearest_cluster.argtypes = [_f32p, sz, sz, _f32p, _u32p, _f32p]
        L.rqo_calculate_recall.restype = C.c_float
        L.rqo_calculate_recall.argtypes = [_i32p, sz, _i32p, sz]
        L.rqo_ord32_from_f32.restype = C.c_int32
        L.rqo_ord32_from_f32.argtypes = [C.c_float]
        L.rqo_ord32_to_f32.restype = C.c_float
        L.rqo_ord32_to_f32.argtypes = [C.c_int32]
        L.rqo_metrics_get.restype = None
        L.rqo_metrics_get.argtypes = [C.POINTER(Metrics)]
        L.rqo_metrics_reset.restype = None
        L.rqo_build.restype = C.POINTER(_Index)
        L.rqo_build.argtypes = [_f32p, C.c_uint64, C.c_uint32, _f32p, C.c_uint32, _f32p]
        L.rqo_free.restype = None
        L.rqo_free.argtypes = [C.POINTER(_Index)]
        L.rqo_dump_dir.restype = C.c_int
        L.rqo_dump_dir.argtypes = [C.POINTER(_Index), C.c_char_p]
        L.rqo_load_dir.restype = C.POINTER(_Index)
        L.rqo_load_dir.argtypes = [C.c_char_p]
        L.rqo_query.restype = C.c_int
        L.rqo_query.argtypes = [C.POINTER(_Index), _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                _f32p, _u32p, _u32p]
        L.rqo_coarse_rank.restype = C.c_int
        L.rqo_coarse_rank.argtypes = [C.POINTER(_Index), _f32p, C.c_uint32, _u32p, _f32p]
        L.rqo_query_prep.restype = None
        L.rqo_query_prep.argtypes = [C.POINTER(_Index), _f32p, C.c_uint32, _f32p, _f32p, _u32p, _u64p]
        L.rqo_scan_cluster.restype = None
        L.rqo_scan_cluster.argtypes = [C.POINTER(_Index), C.c_uint32, C.c_float, _u64p, C.c_float,
                                       C.c_float, C.c_float, _f32p]
        L.rqo_scan_only.restype = C.c_uint64
        L.rqo_scan_only.argtypes = [C.POINTER(_Index), _f32p, C.c_uint32, C.c_uint32, _f32p]
        L.rqo_view.restype = C.POINTER(_Index)
        L.rqo_view.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, _f32p, _f32p, _f32p, _u32p, _u32p,
                               _u64p, _f32p]
        L.rqo_free_view.restype = None
        L.rqo_free_view.argtypes = [C.POINTER(_Index)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- kernel-level wrappers (src/simd.rs, src/utils.rs) -------------------------------------
def l2_squared_distance(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().rqo_l2_squared_distance(_p(a, _f32p), _p(b, _f32p), a.size))


def vector_dot_product(a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(lib().rqo_vector_dot_product(_p(a, _f32p), _p(b, _f32p), a.size))


def min_max_residual(x, y):
    x, y = _f32(x), _f32(y)
    res = np.empty_like(x)
    lo, hi = C.c_float(), C.c_float()
    lib().rqo_min_max_residual(_p(res, _f32p), _p(x, _f32p), _p(y, _f32p), x.size, C.byref(lo), C.byref(hi))
    return res, np.float32(lo.value), np.float32(hi.value)


def scalar_quantize(vec, lower_bound, multiplier):
    vec = _f32(vec)
    q = np.zeros(vec.size, dtype=np.uint8)
    s = lib().rqo_scalar_quantize(_p(q, _u8p), _p(vec, _f32p), vec.size, np.float32(lower_bound),
                                  np.float32(multiplier))
    return q, int(s)


def vector_binarize_query(q):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    out = np.zeros(q.size // 64 * 4, dtype=np.uint64)
    lib().rqo_vector_binarize_query(_p(q, _u8p), q.size, _p(out, _u64p))
    return out


def binary_dot_product(x, y) -> int:
    x = np.ascontiguousarray(x, dtype=np.uint64)
    y = np.ascontiguousarray(y, dtype=np.uint64)
    return int(lib().rqo_binary_dot_product(_p(x, _u64p), _p(y, _u64p), x.size))


def asymmetric_binary_dot_product(x, y) -> int:
    x = np.ascontiguousarray(x, dtype=np.uint64)
    y = np.ascontiguousarray(y, dtype=np.uint64)
    assert y.size == 4 * x.size
    return int(lib().rqo_asymmetric_binary_dot_product(_p(x, _u64p), _p(y, _u64p), x.size))


def vector_binarize_u64(v):
    v = _f32(v)
    out = np.zeros((v.size + 63) // 64, dtype=np.uint64)
    lib().rqo_vector_binarize_u64(_p(v, _f32p), v.size, _p(out, _u64p))
    return out


def project(vec, orthogonal):
    """y = vec^T P in the AVX2 `project` order (utils.rs:237-258). orthogonal is P, row-major."""
    vec = _f32(vec)
    pt = _f32(np.asarray(orthogonal, dtype=np.float32).T)
    out = np.empty_like(vec)
    lib().rqo_project(_p(vec, _f32p), _p(pt, _f32p), vec.size, _p(out, _f32p))
    return out


def project_rows(x, orthogonal):
    x = _f32(x)
    pt = _f32(np.asarray(orthogonal, dtype=np.float32).T)
    out = np.empty_like(x)
    L = lib()
    for i in range(x.shape[0]):
        L.rqo_project(_p(x[i], _f32p), _p(pt, _f32p), x.shape[1], _p(out[i], _f32p))
    return out


def kmeans_nearest_cluster(centroids, vec):
    centroids, vec = _f32(centroids), _f32(vec)
    lab, d = C.c_uint32(), C.c_float()
    lib().rqo_kmeans_nearest_cluster(_p(centroids, _f32p), centroids.shape[0], centroids.shape[1],
                                     _p(vec, _f32p), C.byref(lab), C.byref(d))
    return int(lab.value), np.float32(d.value)


def calculate_recall(truth, res, topk) -> float:
    truth = np.ascontiguousarray(truth, dtype=np.int32)
    res = np.ascontiguousarray(res, dtype=np.int32)
    return float(lib().rqo_calculate_recall(_p(truth, _i32p), truth.size, _p(res, _i32p), topk))


def ord32_from_f32(x) -> int:
    return int(lib().rqo_ord32_from_f32(np.float32(x)))


def ord32_to_f32(k):
    return np.float32(lib().rqo_ord32_to_f32(int(k)))


def metrics() -> dict:
    m = Metrics()
    lib().rqo_metrics_get(C.byref(m))
    return {"rough": m.rough, "precise": m.precise, "query": m.query, "miss": m.miss}


def metrics_reset():
    lib().rqo_metrics_reset()


# ---- index -----------------------------------------------------------------------------------
class OracleIndex:
    """CPU restatement of `RaBitQ` (src/rabitq.rs:57-68)."""

    def __init__(self, ptr, view_refs=None):
        self._ptr = ptr
        self._view_refs = view_refs  # keeps borrowed numpy arrays alive for views

    # RaBitQ::from_path on arrays; `orthogonal` is an input (the reference's RNG is unseeded)
    @classmethod
    def build(cls, base, centroids, orthogonal):
        base, centroids, orthogonal = _f32(base), _f32(centroids), _f32(orthogonal)
        n, d = base.shape
        k = centroids.shape[0]
        dpad = (d + 63) // 64 * 64
        assert centroids.shape[1] == d and orthogonal.shape == (dpad, dpad)
        ptr = lib().rqo_build(_p(base, _f32p), n, d, _p(centroids, _f32p), k, _p(orthogonal, _f32p))
        return cls(ptr)

    @classmethod
    def load_from_dir(cls, path):
        ptr = lib().rqo_load_dir(os.fsencode(path))
        if not ptr:
            raise IOError(f"oracle: cannot load index from {path}")
        return cls(ptr)

    @classmethod
    def view(cls, dim, base, orthogonal, centroids, offsets, map_ids, codes, factors):
        """Borrow arrays (no copy) -- used by the CPU baseline on GPU-built indexes."""
        arrs = [_f32(base), _f32(orthogonal), _f32(centroids), np.ascontiguousarray(offsets, dtype=np.uint32),
                np.ascontiguousarray(map_ids, dtype=np.uint32), np.ascontiguousarray(codes, dtype=np.uint64),
                _f32(factors)]
        n, k = arrs[4].size, arrs[3].size - 1
        ptr = lib().rqo_view(dim, n, k, _p(arrs[0], _f32p), _p(arrs[1], _f32p), _p(arrs[2], _f32p),
                             _p(arrs[3], _u32p), _p(arrs[4], _u32p), _p(arrs[5], _u64p), _p(arrs[6], _f32p))
        return cls(ptr, view_refs=arrs)

    def dump_to_dir(self, path):
        if lib().rqo_dump_dir(self._ptr, os.fsencode(path)) != 0:
            raise IOError(f"oracle: cannot dump index to {path}")

    def close(self):
        if self._ptr:
            (lib().rqo_free_view if self._view_refs is not None else lib().rqo_free)(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # array views (copies)
    @property
    def dim(self):
        return int(self._ptr.contents.dim)

    @property
    def n(self):
        return int(self._ptr.contents.n)

    @property
    def k(self):
        return int(self._ptr.contents.k)

    def _arr(self, field, shape, dtype):
        p = getattr(self._ptr.contents, field)
        count = int(np.prod(shape))
        if count == 0:
            return np.zeros(shape, dtype=dtype)
        return np.ctypeslib.as_array(p, shape=(count,)).view(dtype).reshape(shape).copy()

    @property
    def base(self):
        return self._arr("base", (self.n, self.dim), np.float32)

    @property
    def orthogonal(self):
        return self._arr("orthogonal", (self.dim, self.dim), np.float32)

    @property
    def centroids(self):
        return self._arr("centroids", (self.k, self.dim), np.float32)

    @property
    def offsets(self):
        return self._arr("offsets", (self.k + 1,), np.uint32)

    @property
    def map_ids(self):
        return self._arr("map_ids", (self.n,), np.uint32)

    @property
    def codes(self):
        return self._arr("x_binary_vec", (self.n, self.dim // 64), np.uint64)

    @property
    def factors(self):
        return self._arr("factors", (self.n, 4), np.float32)

    # RaBitQ::query
    def query(self, q, probe, topk, heuristic_rank=False):
        q = _f32(q)
        d = np.empty(max(topk, 1), dtype=np.float32)
        ids = np.empty(max(topk, 1), dtype=np.uint32)
        cnt = C.c_uint32()
        rc = lib().rqo_query(self._ptr, _p(q, _f32p), q.size, probe, topk, int(heuristic_rank),
                             _p(d, _f32p), _p(ids, _u32p), C.byref(cnt))
        if rc != 0:
            raise RuntimeError(f"oracle query failed where the reference panics (code {rc})")
        return d[:cnt.value].copy(), ids[:cnt.value].copy()

    def rotate_query(self, q):
        q = _f32(q)
        qp = np.zeros(self.dim, dtype=np.float32)
        qp[:q.size] = q
        y = np.empty(self.dim, dtype=np.float32)
        lib().rqo_project(_p(qp, _f32p), self._ptr.contents.orthogonal_t, self.dim, _p(y, _f32p))
        return y

    def coarse_rank(self, y, probe):
        y = _f32(y)
        cap = max(min(probe, self.k), 1)
        cl = np.empty(cap, dtype=np.uint32)
        dist = np.empty(cap, dtype=np.float32)
        n = lib().rqo_coarse_rank(self._ptr, _p(y, _f32p), probe, _p(cl, _u32p), _p(dist, _f32p))
        if n < 0:
            raise RuntimeError("probe == 0 (reference panics)")
        return cl[:n].copy(), dist[:n].copy()

    def query_prep(self, y, cluster):
        y = _f32(y)
        lo, delta, s = C.c_float(), C.c_float(), C.c_uint32()
        planes = np.zeros(self.dim // 64 * 4, dtype=np.uint64)
        lib().rqo_query_prep(self._ptr, _p(y, _f32p), cluster, C.byref(lo), C.byref(delta), C.byref(s),
                             _p(planes, _u64p))
        return np.float32(lo.value), np.float32(delta.value), int(s.value), planes

    def scan_cluster(self, cluster, ycd, planes, lower, scalar_sum, delta):
        off = self.offsets
        out = np.empty(int(off[cluster + 1] - off[cluster]), dtype=np.float32)
        planes = np.ascontiguousarray(planes, dtype=np.uint64)
        lib().rqo_scan_cluster(self._ptr, cluster, np.float32(ycd), _p(planes, _u64p), np.float32(lower),
                               np.float32(scalar_sum), np.float32(delta), _p(out, _f32p))
        return out

    def scan_only(self, q, probe, scratch):
        q = _f32(q)
        return int(lib().rqo_scan_only(self._ptr, _p(q, _f32p), q.size, probe, _p(scratch, _f32p)))
