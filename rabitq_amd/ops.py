"""Per-stage entry points of the engine (one hot kernel each), used by the parity tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib
from .index import _addr, _f32


def rotate(x, orthogonal, use_mfma: bool = True):
    """X' = X P in the reference's `project` order (src/utils.rs:237-258)."""
    x, P = _f32(x), _f32(orthogonal)
    out = np.empty_like(x)
    check(lib().rq_rotate(_addr(x), x.shape[0], x.shape[1], _addr(P), int(use_mfma), _addr(out)))
    return out


def quantize_pack(x_rot, centroids_rot):
    """labels, centroid distances, codes, factors for already-rotated vectors (rabitq.rs:203-229)."""
    x, c = _f32(x_rot), _f32(centroids_rot)
    n, dim = x.shape
    label = np.empty(n, np.uint32)
    dist = np.empty(n, np.float32)
    codes = np.empty((n, dim // 64), np.uint64)
    factors = np.empty((n, 4), np.float32)
    check(lib().rq_quantize_pack(_addr(x), n, dim, _addr(c), c.shape[0], _addr(label), _addr(dist), _addr(codes),
                                 _addr(factors)))
    return label, dist, codes, factors


def coarse_rank(index, queries, probe):
    q = _f32(queries)
    nprobe = min(probe, index.k)
    y = np.empty((q.shape[0], index.dim), np.float32)
    cl = np.empty((q.shape[0], nprobe), np.uint32)
    dist = np.empty((q.shape[0], nprobe), np.float32)
    check(lib().rq_coarse_rank(index._h, _addr(q), q.shape[0], q.shape[1], probe, _addr(y), _addr(cl), _addr(dist)))
    return y, cl, dist


def query_prep(index, y, clusters):
    y = _f32(y)
    cl = np.ascontiguousarray(clusters, dtype=np.uint32)
    nq = y.shape[0]
    lo = np.empty(nq, np.float32)
    delta = np.empty(nq, np.float32)
    s = np.empty(nq, np.uint32)
    planes = np.empty((nq, 4 * index.dim // 64), np.uint64)
    check(lib().rq_query_prep(index._h, _addr(y), nq, _addr(cl), _addr(lo), _addr(delta), _addr(s), _addr(planes)))
    return lo, delta, s, planes


def scan(index, cluster, ycd, planes, lower, scalar_sum, delta, list_len):
    planes = np.ascontiguousarray(planes, dtype=np.uint64)
    out = np.empty(list_len, np.float32)
    check(lib().rq_scan(index._h, cluster, C.c_float(ycd), _addr(planes), C.c_float(lower), C.c_float(scalar_sum),
                        C.c_float(delta), _addr(out)))
    return out


def rerank(index, query_padded, positions):
    q = _f32(query_padded)
    pos = np.ascontiguousarray(positions, dtype=np.uint32)
    out = np.empty(pos.size, np.float32)
    check(lib().rq_rerank(index._h, _addr(q), _addr(pos), pos.size, _addr(out)))
    return out


def kmeans_device(base_ptr: int, n: int, d: int, k: int, out_ptr: int, iters: int = 20, points_per_centroid: int = 256,
                  seed: int = 0) -> None:
    """Train k centroids on device-resident vectors (scripts/cluster.py's job, done on the GPU)."""
    check(lib().rq_kmeans_device(C.c_void_p(base_ptr), n, d, k, iters, points_per_centroid, seed, C.c_void_p(out_ptr)))
