"""Second, independent restatement (numpy / pure Python) of the reference's hot path.

TEST INFRASTRUCTURE ONLY (same rules as rabitq_oracle.h).  Parity unpinned by the reference.

Two jobs:
  1. restate the reference's SCALAR fallbacks (`*_raw` in src/utils.rs), which the author keeps
     beside every SIMD kernel as a designed cross-check, so the C/AVX2 oracle can
     be checked against something written separately;
  2. restate the control flow of RaBitQ::from_path / RaBitQ::query / the re-rankers
     (src/rabitq.rs, src/rerank.rs) in plain Python for SMALL cases, including Rust's
     BinaryHeap push/pop, so the C oracle's orchestration is checked too.
All paths cite /root/reference/<file>:<line>.
"""
from __future__ import annotations

import numpy as np

F32_MAX = np.float32(np.finfo(np.float32).max)
THETA_LOG_DIM = 4       # consts.rs:8
EPSILON = np.float32(1.9)  # consts.rs:6
DEFAULT_X_DOT_PRODUCT = np.float32(0.8)  # consts.rs:4
SCALAR = np.float32(1.0) / np.float32(15.0)  # consts.rs:10
WINDOW_SIZE = 12        # consts.rs:12


# ---- src/utils.rs scalar fallbacks ------------------------------------------------------------
def vector_binarize_query_raw(vec: np.ndarray) -> np.ndarray:
    """utils.rs:90-97: binary[(i + j*len)/64] |= ((vec[i] >> j) & 1) << (i % 64)."""
    vec = np.asarray(vec, dtype=np.uint8)
    n = vec.size
    out = np.zeros(n * THETA_LOG_DIM // 64, dtype=np.uint64)
    for j in range(THETA_LOG_DIM):
        for i in range(n):
            bit = (int(vec[i]) >> j) & 1
            out[(i + j * n) // 64] |= np.uint64(bit << (i % 64))
    return out


def binary_dot_product_raw(x: np.ndarray, y: np.ndarray) -> int:
    """utils.rs:101-107."""
    return sum(bin(int(a) & int(b)).count("1") for a, b in zip(x, y))


def asymmetric_binary_dot_product_raw(x: np.ndarray, y: np.ndarray) -> int:
    """utils.rs:113-135."""
    n = len(x)
    return sum(binary_dot_product_raw(x, y[p * n:(p + 1) * n]) << p for p in range(THETA_LOG_DIM))


def min_max_raw(x: np.ndarray, y: np.ndarray):
    """utils.rs:155-168."""
    res = (np.asarray(x, np.float32) - np.asarray(y, np.float32)).astype(np.float32)
    mn, mx = F32_MAX, -F32_MAX
    for v in res:
        if v < mn:
            mn = v
        if v > mx:
            mx = v
    return res, np.float32(mn), np.float32(mx)


def scalar_quantize_raw(vec, bias, lower_bound, multiplier):
    """utils.rs:194-209: floor((v - lo) * mult + bias) as u8 -- NOT the parity target: the AVX2
    path (simd.rs:214-215) rounds to nearest even and ignores the bias."""
    vec = np.asarray(vec, np.float32)
    t = ((vec - np.float32(lower_bound)) * np.float32(multiplier)).astype(np.float32) + np.asarray(bias, np.float32)
    q = np.clip(np.floor(t.astype(np.float32)), 0, 255).astype(np.uint8)  # `as u8` saturates
    return q, int(q.astype(np.uint32).sum())


def scalar_quantize_rne(vec, lower_bound, multiplier):
    """simd.rs:214-215 semantics restated with numpy: RNE((v - lo) * mult), low byte, wrapping i32 sum."""
    vec = np.asarray(vec, np.float32)
    t = ((vec - np.float32(lower_bound)).astype(np.float32) * np.float32(multiplier)).astype(np.float32)
    with np.errstate(invalid="ignore"):
        r = np.rint(t.astype(np.float64))
    bad = ~np.isfinite(t) | (r >= 2147483648.0) | (r < -2147483648.0)
    q32 = np.where(bad, -2147483648, np.where(bad, 0, r)).astype(np.int64)  # cvtps_epi32 "indefinite"
    q = (q32 & 0xFF).astype(np.uint8)
    s = int(q32.sum()) & 0xFFFFFFFF
    return q, s


def vector_binarize_u64(vec) -> np.ndarray:
    """utils.rs:53-61: bit i set iff vec[i] > 0.0."""
    vec = np.asarray(vec, np.float32)
    out = np.zeros((vec.size + 63) // 64, dtype=np.uint64)
    for i, v in enumerate(vec):
        if v > 0.0:
            out[i // 64] |= np.uint64(1 << (i % 64))
    return out


# ---- src/ord32.rs:12-26 -----------------------------------------------------------------------
def ord32_from_f32(x) -> int:
    bits = int(np.float32(x).view(np.int32))
    mask = ((bits >> 31) & 0xFFFFFFFF) >> 1
    r = (bits ^ mask) & 0xFFFFFFFF
    return r - (1 << 32) if r & 0x80000000 else r


def ord32_to_f32(key: int):
    mask = ((key >> 31) & 0xFFFFFFFF) >> 1
    bits = (key ^ mask) & 0xFFFFFFFF
    return np.uint32(bits).view(np.float32)


# ---- AVX2 lane-order emulation of simd.rs:14-73 / :257-314 ------------------------------------
def _fma32(a, b, c):
    """f32 fused multiply-add on arrays: a*b is exact in f64; the f64 add is rounded once more when
    narrowed, so a result can differ from a true FMA in a rare double-rounding tie (tests allow it)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _reduce8(acc):
    c = [np.float32(acc[i] + acc[i + 4]) for i in range(4)]
    return np.float32(np.float32(c[0] + c[1]) + np.float32(c[2] + c[3]))


def l2_squared_distance_lanes(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    acc = np.zeros(8, np.float32)
    for c in range(a.size // 8):
        d = (a[8 * c:8 * c + 8] - b[8 * c:8 * c + 8]).astype(np.float32)
        acc = _fma32(d, d, acc)
    res = _reduce8(acc)
    for i in range(a.size // 8 * 8, a.size):
        r = np.float32(a[i] - b[i])
        res = np.float32(res + np.float32(r * r))
    return res


def vector_dot_product_lanes(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    acc = np.zeros(8, np.float32)
    for c in range(a.size // 8):
        acc = _fma32(a[8 * c:8 * c + 8], b[8 * c:8 * c + 8], acc)
    res = _reduce8(acc)
    for i in range(a.size // 8 * 8, a.size):
        res = np.float32(res + np.float32(a[i] * b[i]))
    return res


# ---- Rust std BinaryHeap (max-heap), keys only (ord32.rs:43-66 makes ids compare Equal) -------
class RustBinaryHeap:
    def __init__(self):
        self.data = []  # list of (key, id)

    def _sift_up(self, start, pos):
        hole = self.data[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if hole[0] <= self.data[parent][0]:
                break
            self.data[pos] = self.data[parent]
            pos = parent
        self.data[pos] = hole

    def push(self, item):
        self.data.append(item)
        self._sift_up(0, len(self.data) - 1)

    def pop(self):
        item = self.data.pop()
        if self.data:
            item, self.data[0] = self.data[0], item
            end, pos = len(self.data), 0
            hole = self.data[0]
            child = 1
            while child + 1 < end:  # child <= end.saturating_sub(2)
                if self.data[child][0] <= self.data[child + 1][0]:
                    child += 1
                self.data[pos] = self.data[child]
                pos = child
                child = 2 * pos + 1
            if child == end - 1:
                self.data[pos] = self.data[child]
                pos = child
            self.data[pos] = hole
            self._sift_up(0, pos)
        return item


# ---- src/rabitq.rs / src/rerank.rs control flow, small cases ----------------------------------
class PyRaBitQ:
    """Plain-Python model of RaBitQ::from_path + RaBitQ::query.  `l2` and `dot` are injectable so a
    test can run the control flow on the C oracle's exact-order float kernels."""

    def __init__(self, base, centroids, orthogonal, l2=l2_squared_distance_lanes,
                 dot=vector_dot_product_lanes):
        self.l2, self.dot = l2, dot
        base = np.asarray(base, np.float32)
        centroids = np.asarray(centroids, np.float32)
        n, d = base.shape
        k = centroids.shape[0]
        dim = (d + 63) // 64 * 64                                     # rabitq.rs:168-179
        b = np.zeros((n, dim), np.float32)
        b[:, :d] = base
        c = np.zeros((k, dim), np.float32)
        c[:, :d] = centroids
        P = np.asarray(orthogonal, np.float32)
        self.dim, self.n, self.k, self.P = dim, n, k, P
        PT = np.ascontiguousarray(P.T)
        xp = np.array([[dot(b[i], PT[j]) for j in range(dim)] for i in range(n)], np.float32).reshape(n, dim)
        self.centroids = np.array([[dot(c[i], PT[j]) for j in range(dim)] for i in range(k)], np.float32).reshape(k, dim)
        dim_sqrt = np.sqrt(np.float32(dim))
        labels = [[] for _ in range(k)]
        factors = np.zeros((n, 4), np.float32)
        codes = np.zeros((n, dim // 64), np.uint64)
        x_c_distance = np.zeros(n, np.float32)
        x_dot = np.zeros(n, np.float32)
        sign_sum = np.zeros(n, np.float32)
        for i in range(n):                                             # rabitq.rs:199-216
            best, lab = F32_MAX, 0
            for j in range(k):                                         # utils.rs:261-277
                dist = l2(self.centroids[j], xp[i])
                if dist < best:
                    best, lab = dist, j
            labels[lab].append((i, best))
            r = (xp[i] - self.centroids[lab]).astype(np.float32)
            x_c_distance[i] = np.sqrt(np.float32(l2(xp[i], self.centroids[lab])))
            factors[i, 3] = x_c_distance[i] * x_c_distance[i]
            codes[i] = vector_binarize_u64(r)
            sgn = np.where(r > 0, np.float32(1), np.float32(-1)).astype(np.float32)
            sign_sum[i] = sgn.sum(dtype=np.float64)
            norm = np.float32(x_c_distance[i] * dim_sqrt)
            is_normal = np.isfinite(norm) and abs(norm) >= np.finfo(np.float32).tiny
            x_dot[i] = np.float32(dot(r, sgn)) / norm if is_normal else DEFAULT_X_DOT_PRODUCT
        error_base = np.float32(2.0) * EPSILON / np.sqrt(np.float32(dim) - np.float32(1.0))
        with np.errstate(all="ignore"):
            for i in range(n):                                         # rabitq.rs:220-229
                over = np.float32(x_c_distance[i] / x_dot[i])
                factors[i, 2] = error_base * np.sqrt(np.float32(over * over - factors[i, 3]))
                factors[i, 0] = np.float32(np.float32(-2.0) / dim_sqrt) * over
                factors[i, 1] = factors[i, 0] * sign_sum[i]
        flat = []
        self.offsets = np.zeros(k + 1, np.uint32)
        for j in range(k):                                             # rabitq.rs:232-243 (stable)
            lst = sorted(labels[j], key=lambda t: ord32_from_f32(t[1]))
            flat += [i for i, _ in lst]
            self.offsets[j + 1] = self.offsets[j] + len(lst)
        self.map_ids = np.array(flat, np.uint32).reshape(-1)
        self.base = b[self.map_ids] if n else b
        self.codes = codes[self.map_ids] if n else codes
        self.factors = factors[self.map_ids] if n else factors

    def query(self, query, probe, topk, heuristic_rank=False):
        q = np.zeros(self.dim, np.float32)
        query = np.asarray(query, np.float32)
        assert self.dim == (query.size + 63) // 64 * 64                # rabitq.rs:275
        q[:query.size] = query
        PT = np.ascontiguousarray(self.P.T)
        y = np.array([self.dot(q, PT[j]) for j in range(self.dim)], np.float32)
        lists = sorted(((ord32_from_f32(self.l2(self.centroids[i], y)), i) for i in range(self.k)))
        lists = lists[:min(probe, self.k)]                             # rabitq.rs:294-297
        threshold = F32_MAX
        heap = RustBinaryHeap()
        arr, count, recent = [], 0, -F32_MAX
        rough_n = precise_n = 0
        W = self.dim // 64
        for key, c in lists:
            ycd = ord32_to_f32(key)
            res, lo, hi = min_max_raw(y, self.centroids[c])
            delta = np.float32(np.float32(hi - lo) * SCALAR)
            with np.errstate(all="ignore"):
                one_over = np.float32(1.0) / delta
                qz, sumq = scalar_quantize_rne(res, lo, one_over)
            planes = vector_binarize_query_raw(qz)
            ds = np.sqrt(np.float32(ycd))
            for j in range(int(self.offsets[c]), int(self.offsets[c + 1])):
                f = self.factors[j]
                s = np.float32(asymmetric_binary_dot_product_raw(self.codes[j], planes))
                with np.errstate(all="ignore"):
                    t = np.float32(f[3] + ycd)
                    t = np.float32(t + np.float32(lo * f[1]))
                    u = np.float32(np.float32(np.float32(np.float32(2.0) * s) - np.float32(sumq)) * f[0])
                    t = np.float32(t + np.float32(u * delta))
                    rough = np.float32(t - np.float32(f[2] * ds))
                rough_n += 1
                if not rough < threshold:                              # rerank.rs:84 / :146
                    continue
                acc = np.float32(self.l2(self.base[j], q))
                precise_n += 1
                if not acc < threshold:
                    continue
                if not heuristic_rank:                                 # rerank.rs:93-100
                    heap.push((ord32_from_f32(acc), int(self.map_ids[j])))
                    if len(heap.data) > topk:
                        heap.pop()
                    if len(heap.data) == topk:
                        threshold = ord32_to_f32(heap.data[0][0])
                else:                                                  # rerank.rs:155-162
                    arr.append((acc, int(self.map_ids[j])))
                    count += 1
                    recent = max(recent, acc)
                    if count >= WINDOW_SIZE:
                        threshold, count, recent = recent, 0, -F32_MAX
        if not heuristic_rank:
            out = [(ord32_to_f32(kk), i) for kk, i in heap.data]
        else:
            order = sorted(range(len(arr)), key=lambda t: (ord32_from_f32(arr[t][0]), t))[:topk]
            out = [arr[t] for t in order]
        return out, {"rough": rough_n, "precise": precise_n}
