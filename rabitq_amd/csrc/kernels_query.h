// kernels_query.h -- gfx950 kernels for RaBitQ::query (src/rabitq.rs:268-367, src/rerank.rs).
//
// Mapping (MI355X-first, not a translation of the reference's per-vector SIMD loop):
//   * candidates live one-per-lane in VGPRs (code words + Factor), queries stream through the
//     scalar unit: every per-query operand (bit planes, lower/delta/sum, threshold) is
//     wave-uniform, so it is fetched with s_load and fed to VALU ops as an SGPR operand.  A list is
//     read from HBM once per launch and scored against every query of the batch that probes it.
//   * the scan filters against the per-query re-rank threshold in-kernel and emits only
//     survivors; exact f32 distances are computed for survivors only; an ordered replay of the
//     reference's heap logic over (rough, accurate) pairs reproduces its result id-for-id.
#pragma once
#ifndef RQ_COLLECT_UNROLL
#define RQ_COLLECT_UNROLL 8  // 16 tiles per step: coarse 1.07 -> 1.03 ms per step (4: the round-4 form; 16 no better)
#endif
#include "common.h"
#include "kernels_scan_common.h"

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// zero-pad queries to the padded dimension (src/rabitq.rs:277-280)
// ------------------------------------------------------------------------------------------------
__global__ void pad_rows_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t n,
                                uint32_t len, uint32_t dim) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    uint64_t r = i / dim;
    uint32_t c = (uint32_t)(i - r * dim);
    out[i] = c < len ? in[r * len + c] : 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Rotation, VALU form: out[r][j] = vector_dot_product(x[r], P[:, j]) in the exact AVX2 order
// (src/utils.rs:237-258 -> src/simd.rs:257-314): 8 accumulators acc[l] = fma(x[8c+l], P[8c+l][j],
// acc[l]) over chunks c, then the fixed fold.  block (64,4): lane <-> column j (coalesced P reads),
// x[r][*] is a wave-uniform broadcast.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rotate_valu_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ P,
                                                          float *__restrict__ out, uint64_t n,
                                                          uint32_t dim) {
    uint64_t r = (uint64_t)blockIdx.x * 4 + threadIdx.y;
    uint32_t j = blockIdx.y * 64 + threadIdx.x;
    if (r >= n) return;
    const float *xr = x + r * dim;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t c = 0; c < dim; c += 8) {
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[l] = fmaf(xr[c + l], P[(uint64_t)(c + l) * dim + j], acc[l]);
    }
    out[r * dim + j] = reduce8_regs(acc);
}

// ------------------------------------------------------------------------------------------------
// Coarse distances (src/rabitq.rs:285-293): dist[q][j] = l2_squared_distance(centroid_j, y_q) in
// the exact order of src/simd.rs:14-73 (diff rounded, then fused square-accumulate per AVX lane).
// lane <-> centroid j over the TRANSPOSED rotated centroids cent_t[dim][k] (coalesced; this is also
// the reference's on-disk centroids.fvecs layout), QT queries per thread held in LDS (broadcast).
// ------------------------------------------------------------------------------------------------
// Large batches: the same distances with the QUERY side in scalar registers.  A lane still owns one list; the block's
// QT queries are read through the scalar unit (their rows are wave-uniform), 16 dimensions of one query per
// s_load_dwordx16, and enter the packed ops as SGPR pairs: two neighbouring dimensions (= two neighbouring AVX
// lanes of src/simd.rs:14-73, each with its own accumulator and per-component rounding) per v_pk_add_f32 /
// v_pk_fma_f32.  No LDS: the LDS-broadcast form above spends as many LDS cycles as VALU cycles per element and
// stalls at half the packed-f32 rate.  QT queries per centroid element loaded (8: 86 VGPRs, five waves per SIMD; 16 was
// measured slower, three waves per SIMD do not cover the scalar loads).
template <int QT>
__global__ __launch_bounds__(256) void coarse_dist_sreg_kernel(const float *__restrict__ cent_t,
                                                               const float *__restrict__ y, float *__restrict__ dist,
                                                               uint32_t k, uint32_t dim, uint32_t nq, uint32_t kstride) {
    const uint32_t q0 = blockIdx.x * QT;
    const uint32_t j = blockIdx.y * 256 + threadIdx.x;
    const bool live = j < k;
    const float *cp = cent_t + (live ? j : 0);
    f32x2 acc[QT][4];  // [query][pair of AVX lanes]
#pragma unroll
    for (int v = 0; v < QT; ++v)
#pragma unroll
        for (int l = 0; l < 4; ++l) acc[v][l] = f32x2{0.0f, 0.0f};
    static_assert(QT % 4 == 0, "queries are fetched four at a time");
    for (uint32_t c = 0; c < dim; c += 16) {  // dim is a multiple of 64
        float ce[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) ce[l] = cp[(uint64_t)(c + l) * kstride];  // 16 loads in flight
#pragma unroll
        for (int v0 = 0; v0 < QT; v0 += 4) {
            float yv[4][16];  // four queries x 16 dimensions: four s_load_dwordx16 issued together
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const uint32_t q = q0 + v0 + v < nq ? q0 + v0 + v : nq - 1;  // uniform; rows past the batch are computed and dropped
                const float *yq = y + (uint64_t)q * dim + c;
#pragma unroll
                for (int l = 0; l < 16; ++l) yv[v][l] = yq[l];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)  // the two 8-dimension steps of the chunk, in order (one accumulator chain per AVX lane)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int l = 0; l < 4; ++l) {
                        const f32x2 c2 = {ce[8 * h + 2 * l], ce[8 * h + 2 * l + 1]};
                        const f32x2 y2 = {yv[v][8 * h + 2 * l], yv[v][8 * h + 2 * l + 1]};
                        const f32x2 d2 = c2 - y2;
                        acc[v0 + v][l] = __builtin_elementwise_fma(d2, d2, acc[v0 + v][l]);
                    }
        }
    }
    if (live) {
#pragma unroll
        for (int v = 0; v < QT; ++v) {
            float a[8];
#pragma unroll
            for (int l = 0; l < 4; ++l) a[2 * l] = acc[v][l].x, a[2 * l + 1] = acc[v][l].y;
            if (q0 + v < nq) dist[(uint64_t)(q0 + v) * k + j] = reduce8_regs(a);
        }
    }
}

template <int QT>
__global__ __launch_bounds__(256) void coarse_dist_kernel(const float *__restrict__ cent_t,
                                                          const float *__restrict__ y,
                                                          float *__restrict__ dist, uint32_t k,
                                                          uint32_t dim, uint32_t nq, uint32_t kstride) {
    // cent_t points at the first list of the range; k = number of lists ranked, kstride = row stride
    extern __shared__ __attribute__((aligned(16))) float ys[];  // [dim][QT]: one ds_read_b128 = 4 queries at one dimension
    const uint32_t q0 = blockIdx.x * QT;
    const uint32_t j = blockIdx.y * 256 + threadIdx.x;
    for (uint32_t i = threadIdx.x; i < QT * dim; i += 256) {
        uint32_t v = i / dim, e = i - v * dim;  // coalesced reads of y, transposed into LDS
        ys[e * QT + v] = (q0 + v < nq) ? y[(uint64_t)(q0 + v) * dim + e] : 0.0f;
    }
    __syncthreads();
    static_assert(QT % 4 == 0, "queries are processed in packed pairs, read four at a time");
    f32x2 acc[QT / 2][8];  // [query pair][AVX lane]: v_pk_add_f32 + v_pk_fma_f32, per-component rounding
#pragma unroll
    for (int v = 0; v < QT / 2; ++v)
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[v][l] = f32x2{0.0f, 0.0f};
    const bool live = j < k;
    const float *cp = cent_t + (live ? j : 0);
    for (uint32_t c = 0; c < dim; c += 8) {
        float ce[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) ce[l] = cp[(uint64_t)(c + l) * kstride];  // 8 loads in flight
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const f32x2 ce2 = {ce[l], ce[l]};
#pragma unroll
            for (int v4 = 0; v4 < QT / 4; ++v4) {
                const float4 yq = *reinterpret_cast<const float4 *>(&ys[(c + l) * QT + 4 * v4]);
                const f32x2 y01 = {yq.x, yq.y}, y23 = {yq.z, yq.w};
                const f32x2 d01 = ce2 - y01, d23 = ce2 - y23;
                acc[2 * v4][l] = __builtin_elementwise_fma(d01, d01, acc[2 * v4][l]);
                acc[2 * v4 + 1][l] = __builtin_elementwise_fma(d23, d23, acc[2 * v4 + 1][l]);
            }
        }
    }
    if (live) {
#pragma unroll
        for (int v = 0; v < QT / 2; ++v) {
            float a0[8], a1[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) a0[l] = acc[v][l].x, a1[l] = acc[v][l].y;
            if (q0 + 2 * v < nq) dist[(uint64_t)(q0 + 2 * v) * k + j] = reduce8_regs(a0);
            if (q0 + 2 * v + 1 < nq) dist[(uint64_t)(q0 + 2 * v + 1) * k + j] = reduce8_regs(a1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bitonic sort (flip / disperse form: every compare-exchange puts the smaller element at the lower
// index, so elements at index >= n can be treated as +inf and are never touched).
// ------------------------------------------------------------------------------------------------
template <typename T, typename KeyFn>
__device__ __forceinline__ void bitonic_sort_block(T *a, uint32_t n, KeyFn key) {
    if (n < 2) return;
    uint32_t p2 = 1;
    while (p2 < n) p2 <<= 1;
    const uint32_t half = p2 >> 1;
    for (uint32_t k = 2; k <= p2; k <<= 1) {
        // flip
        for (uint32_t t = threadIdx.x; t < half; t += blockDim.x) {
            uint32_t hk = k >> 1;
            uint32_t blk = t / hk, off = t - blk * hk;
            uint32_t i = blk * k + off, j = blk * k + k - 1 - off;
            if (j < n) {
                T ai = a[i], aj = a[j];
                if (key(aj) < key(ai)) {
                    a[i] = aj;
                    a[j] = ai;
                }
            }
        }
        __syncthreads();
        for (uint32_t s = k >> 2; s >= 1; s >>= 1) {
            for (uint32_t t = threadIdx.x; t < half; t += blockDim.x) {
                uint32_t i = (t / s) * (2 * s) + (t % s), j = i + s;
                if (j < n) {
                    T ai = a[i], aj = a[j];
                    if (key(aj) < key(ai)) {
                        a[i] = aj;
                        a[j] = ai;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Probe selection (src/rabitq.rs:294-297): the `nprobe` smallest (distance, cluster id) pairs in
// ascending order.  total_cmp order == Ord32 order; the composite u64 (biased Ord32 << 32 | id) is
// unique, so an 8-bit-per-pass radix select finds the nprobe-th key exactly, then the selected keys
// are bitonic-sorted in LDS.  Exactly-equal distances are ordered by cluster id (the reference's
// select_nth_unstable leaves that order unspecified).  One 256-thread block per query.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_probe_kernel(const float *__restrict__ dist, uint32_t k,
                                                           uint32_t nprobe,
                                                           uint32_t *__restrict__ out_cluster,
                                                           float *__restrict__ out_dist, uint32_t id_offset,
                                                           uint32_t out_stride, const uint32_t *__restrict__ only_rows = nullptr /* per row: 0 = skip */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(smem_raw);  // nprobe entries
    if (only_rows && only_rows[blockIdx.x] == 0u) return;
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_sel, s_want, s_done, s_cnt;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const float *d = dist + (uint64_t)b * k;

    uint64_t prefix = 0, mask = 0, T = ~0ull;
    uint32_t want = nprobe;  // rank (1-based) of the wanted key inside the current prefix group
    bool done = false;
    for (int pass = 7; pass >= 0 && !done; --pass) {
        const int shift = pass * 8;
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t j = tid; j < k; j += 256) {
            uint64_t key = ((uint64_t)ord32_biased(d[j]) << 32) | j;
            if ((key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // wave 0: locate the bin holding rank `want`
            uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            uint32_t s = h0 + h1 + h2 + h3, incl = s;
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t up = __shfl_up(incl, o, 64);
                if ((int)tid >= o) incl += up;
            }
            uint32_t excl = incl - s;
            if (excl < want && want <= incl) {
                uint32_t r = want - excl, sel, cntbin;
                if (r <= h0) { sel = 0; cntbin = h0; }
                else if (r <= h0 + h1) { sel = 1; r -= h0; cntbin = h1; }
                else if (r <= h0 + h1 + h2) { sel = 2; r -= h0 + h1; cntbin = h2; }
                else { sel = 3; r -= h0 + h1 + h2; cntbin = h3; }
                s_sel = 4 * tid + sel;
                s_want = r;
                s_done = (cntbin == r) ? 1u : 0u;  // the whole bin is taken: lower bits don't matter
            }
        }
        __syncthreads();
        prefix |= (uint64_t)s_sel << shift;
        mask |= 0xFFull << shift;
        want = s_want;
        if (s_done) {
            T = prefix | ~mask;
            done = true;
        }
        __syncthreads();
    }
    if (!done) T = prefix;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (uint32_t j = tid; j < k; j += 256) {
        uint64_t key = ((uint64_t)ord32_biased(d[j]) << 32) | j;
        if (key <= T) {
            uint32_t p = atomicAdd(&s_cnt, 1u);
            if (p < nprobe) keys[p] = key;
        }
    }
    __syncthreads();
    bitonic_sort_block(keys, nprobe, [](uint64_t v) { return v; });
    for (uint32_t i = tid; i < nprobe; i += 256) {
        uint64_t key = keys[i];
        out_cluster[(uint64_t)b * out_stride + i] = (uint32_t)key + id_offset;
        out_dist[(uint64_t)b * out_stride + i] = ord32_unbias((uint32_t)(key >> 32));
    }
    for (uint32_t i = nprobe + tid; i < out_stride; i += 256) {  // fewer lists than requested: "no list"
        out_cluster[(uint64_t)b * out_stride + i] = 0xFFFFFFFFu;
        out_dist[(uint64_t)b * out_stride + i] = __builtin_inff();
    }
}

// The same selection with ONE WAVE per query, for nprobe <= 64 and k <= 64*KPL: the row of distances lives
// in registers (lane l holds lists l, l+64, ...), the nprobe-th smallest key is found by bisection on
// the monotone u32 image of the distance (count = per-lane compares + one wave reduction, typically
// ~20 steps, stopping as soon as a threshold selects exactly nprobe), ties at the threshold are broken
// by list id (a second bisection, rare), the winners are compacted by ballot and sorted across the 64
// lanes with a shuffle bitonic network.  No LDS atomics, no block barriers.
// One wave's selection for query row b; `win`: 64 u64 of LDS owned by the calling wave.
template <int KPL>
__device__ __forceinline__ void select_probe_wave(const float *__restrict__ dist, uint32_t k, uint32_t nprobe,
                                                  uint32_t *__restrict__ out_cluster, float *__restrict__ out_dist,
                                                  uint32_t id_offset, uint32_t out_stride, uint32_t b,
                                                  unsigned long long *win) {
    const uint32_t lane = threadIdx.x & 63;
    const float *d = dist + (uint64_t)b * k;
    uint32_t key[KPL];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    // register i of lane l holds list list_of(i) = 256 (i / 4) + 4 l + (i % 4): the row is read 16 bytes per lane, 1 KiB
    // per wave instruction (rows are 16-byte aligned whenever k % 4 == 0; else element by element)
    auto list_of = [&](int i) { return 256u * (uint32_t)(i >> 2) + 4u * lane + (uint32_t)(i & 3); };
    const bool vec4 = (k & 3u) == 0u;
#pragma unroll
    for (int i4 = 0; i4 < KPL; i4 += 4) {
        const uint32_t j0 = list_of(i4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec4 && j0 < k) {
            v = *reinterpret_cast<const float4 *>(d + j0);
        } else if (!vec4) {
            if (j0 < k) v.x = d[j0];
            if (j0 + 1 < k) v.y = d[j0 + 1];
            if (j0 + 2 < k) v.z = d[j0 + 2];
            if (j0 + 3 < k) v.w = d[j0 + 3];
        }
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i4 + e;
            key[i] = 0xFFFFFFFFu;  // "no list": above every real key (a NaN distance with all-ones payload is not supported)
            if (j0 + e < k) {
                key[i] = ord32_biased(ve[e]);
                kmin = key[i] < kmin ? key[i] : kmin;
                kmax = key[i] > kmax ? key[i] : kmax;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t a = __shfl_xor(kmin, o, 64), c = __shfl_xor(kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = c > kmax ? c : kmax;
    }
    // wave-wide count of keys <= t: one compare per register, the lane counts come out of the scalar unit
    // (ballot + s_bcnt1), no cross-lane shuffles in the bisection loop
    auto count_le = [&](uint32_t t) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) c += (uint32_t)__popcll(__ballot(key[i] <= t));
        return c;
    };
    // smallest T with count(key <= T) >= nprobe (nprobe <= k, so T <= kmax)
    // (kmin / kmax are the same in every lane after the butterfly: said explicitly, the bisection runs on scalar registers)
    kmin = __builtin_amdgcn_readfirstlane(kmin), kmax = __builtin_amdgcn_readfirstlane(kmax);
    uint32_t lo = kmin, hi = kmax, T = kmax;
    bool exact = false;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t c = count_le(mid);
        if (c == nprobe) {
            T = mid;
            exact = true;
            break;
        }
        if (c > nprobe) hi = mid;
        else lo = mid + 1;
    }
    if (!exact) T = lo;
    uint32_t J = 0xFFFFFFFFu;  // among keys == T only ids <= J are taken
    if (!exact) {
        const uint32_t c_le = count_le(T);
        if (c_le > nprobe) {  // ties at the threshold: the smallest list ids win
            const uint32_t c_lt = T ? count_le(T - 1) : 0u;
            const uint32_t need = nprobe - c_lt;  // >= 1
            uint32_t jl = 0, jh = k - 1;
            while (jl < jh) {
                const uint32_t jm = jl + ((jh - jl) >> 1);
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < KPL; ++i) c += (key[i] == T && list_of(i) <= jm) ? 1u : 0u;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= need) jh = jm;
                else jl = jm + 1;
            }
            J = jl;
        }
    }
    // compact the nprobe winners (key, id) into LDS
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const uint32_t j = list_of(i);
        const bool take = key[i] < T || (key[i] == T && j <= J && j < k);
        const uint64_t m = __ballot(take);
        if (m) {  // wave-uniform
            if (take) win[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)key[i] << 32) | j;
            base += (uint32_t)__popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    unsigned long long v = lane < nprobe ? win[lane] : ~0ull;
    // bitonic sort across the 64 lanes, ascending
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            const unsigned long long other = __shfl_xor(v, stride, 64);
            const bool up = (lane & size) == 0 || size == 64;
            const bool lower = (lane & stride) == 0;
            const bool take_min = lower == up;
            const unsigned long long mn = other < v ? other : v, mx = other < v ? v : other;
            v = take_min ? mn : mx;
        }
    if (lane < nprobe) {
        out_cluster[(uint64_t)b * out_stride + lane] = (uint32_t)v + id_offset;
        out_dist[(uint64_t)b * out_stride + lane] = ord32_unbias((uint32_t)(v >> 32));
    }
    for (uint32_t i = nprobe + lane; i < out_stride; i += 64) {  // fewer lists than requested: "no list"
        out_cluster[(uint64_t)b * out_stride + i] = 0xFFFFFFFFu;
        out_dist[(uint64_t)b * out_stride + i] = __builtin_inff();
    }
}
template <int KPL>
__global__ __launch_bounds__(256) void select_probe_wave_kernel(const float *__restrict__ dist, uint32_t k,
                                                                uint32_t nprobe, uint32_t *__restrict__ out_cluster,
                                                                float *__restrict__ out_dist, uint32_t id_offset,
                                                                uint32_t out_stride, uint32_t nq) {
    __shared__ unsigned long long win[4][64];
    const uint32_t wave = threadIdx.x >> 6, b = blockIdx.x * 4 + wave;
    if (b >= nq) return;
    select_probe_wave<KPL>(dist, k, nprobe, out_cluster, out_dist, id_offset, out_stride, b, win[wave]);
}

// ------------------------------------------------------------------------------------------------
// Probe selection behind the matrix-core pre-filter of the coarse ranking (coarse_approx_kernel, kernels_build.h): `dist` holds
// APPROXIMATE values a'_j = |c_j|^2 - 2 <c~_j, y~> (bf16 operands), each within m_y of e_j - |y|^2 (e_j = the reference's exact-order
// f32 distance).  One wave per query:
//   1. tau = the nprobe-th smallest a' of the row (the bisection of select_probe_wave);
//   2. candidates = the lists with a' <= tau + 2 m_y: the true nprobe nearest -- and every exact tie with the nprobe-th -- are among
//      them (kernels_build.h has the argument); typically nprobe + a few dozen;
//   3. their EXACT distances in the reference's lane order (src/simd.rs:14-73: 8 GPU lanes = the 8 AVX lanes of one list, folded by
//      reduce8_lanes), 8 lists per wave step;
//   4. the nprobe smallest (distance, list id) keys of the candidates, ascending: exactly what select_probe_wave returns from a row of
//      exact distances.
// A row with more than RQ_COARSE_CAND candidates (near-equidistant centroids) or a margin that is not finite (NaN / inf input) is
// ranked the plain way instead: the wave recomputes ALL k distances in exact order into the row and runs select_probe_wave on it.
// ------------------------------------------------------------------------------------------------
#define RQ_COARSE_CAND 256u
// exact-order distance (src/simd.rs:14-73) of four lists per 8-lane group to the query row yr: lane al of a group carries AVX
// lane al; all 8 lanes of the group return the folded sum.  dim is a multiple of 64: eight AVX steps at a time, all the loads
// of a chunk in flight before the first is used (one load per step, as a plain loop compiles to, made the caller a chain of
// 256 dependent L2 round trips per query).
__device__ __forceinline__ void coarse_exact_dist4(const float *__restrict__ centroids, const float *__restrict__ yr, uint32_t dim,
                                                   uint32_t al, const uint32_t (&jj)[4], float (&ee)[4]) {
    const float *cp[4];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) cp[q] = centroids + (uint64_t)jj[q] * dim + al;
    for (uint32_t e0 = 0; e0 < dim; e0 += 64) {
        float vv[4][8], yv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            yv[i] = yr[e0 + 8 * i + al];
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[q][i] = cp[q][e0 + 8 * i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float df = vv[q][i] - yv[i];
                acc[q] = fmaf(df, df, acc[q]);
            }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ee[q] = reduce8_lanes(acc[q]);
}

// The candidate lists wn[0 .. c2) (list ids; c2 <= RQ_COARSE_CAND, >= nprobe) of query row b: exact keys in the reference's lane
// order, bitonic sort of the RQ_COARSE_CAND slots across the wave, the nprobe smallest written out in ascending order.
__device__ __forceinline__ void coarse_refine_tail(unsigned long long *wn, uint32_t c2, const float *__restrict__ centroids,
                                                   const float *__restrict__ yr, uint32_t dim, uint32_t nprobe, uint32_t b,
                                                   uint32_t *__restrict__ out_cluster, float *__restrict__ out_dist, uint32_t out_stride) {
    const uint32_t lane = threadIdx.x & 63, grp = lane >> 3, al = lane & 7;
    // exact keys, 32 candidates per step
    for (uint32_t c0 = 0; c0 < c2; c0 += 32) {
        uint32_t jj[4];
        float ee[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) jj[q] = c0 + 8 * q + grp < c2 ? (uint32_t)wn[c0 + 8 * q + grp] : 0u;
        coarse_exact_dist4(centroids, yr, dim, al, jj, ee);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c0 + 8 * q + grp < c2 && al == 0) wn[c0 + 8 * q + grp] = ((unsigned long long)ord32_biased(ee[q]) << 32) | jj[q];
    }
    for (uint32_t i = c2 + lane; i < RQ_COARSE_CAND; i += 64) wn[i] = ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // bitonic sort of the RQ_COARSE_CAND slots, ascending: element index i = lane + 64 s (s = the lane's slot); at a step (size, stride)
    // element i keeps the minimum of (i, i ^ stride) iff ((i & stride) == 0) == ((i & size) == 0)  (the last size ascends everywhere)
    constexpr int NS = RQ_COARSE_CAND / 64;
    unsigned long long vs[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) vs[sl] = wn[lane + 64 * sl];
#pragma unroll
    for (int size = 2; size <= 64 * NS; size <<= 1)
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            if (stride >= 64) {  // the partner is another slot of the same lane
                const int ss = stride / 64;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    if (sl & ss) continue;
                    const bool asc = size >= 64 * NS || ((64 * sl) & size) == 0;
                    const unsigned long long a0 = vs[sl], b0 = vs[sl | ss];
                    const unsigned long long mn = a0 < b0 ? a0 : b0, mxv = a0 < b0 ? b0 : a0;
                    vs[sl] = asc ? mn : mxv, vs[sl | ss] = asc ? mxv : mn;
                }
            } else {
                const bool lower = (lane & stride) == 0;
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const unsigned long long o = __shfl_xor(vs[sl], stride, 64);
                    const bool asc = size < 64 ? (lane & size) == 0 : (size >= 64 * NS || ((64 * sl) & size) == 0);
                    const unsigned long long mn = o < vs[sl] ? o : vs[sl], mxv = o < vs[sl] ? vs[sl] : o;
                    vs[sl] = lower == asc ? mn : mxv;
                }
            }
        }
    const unsigned long long v0 = vs[0];
    // v0 of lane l = the l-th smallest key
    if (lane < nprobe) {
        out_cluster[(uint64_t)b * out_stride + lane] = (uint32_t)v0;
        out_dist[(uint64_t)b * out_stride + lane] = ord32_unbias((uint32_t)(v0 >> 32));
    }
    for (uint32_t i = nprobe + lane; i < out_stride; i += 64) {  // fewer lists than requested: "no list"
        out_cluster[(uint64_t)b * out_stride + i] = 0xFFFFFFFFu;
        out_dist[(uint64_t)b * out_stride + i] = __builtin_inff();
    }
}

// the query's margin of the matrix-core pre-filter (kernels_build.h) -> the biased key of tau + 2 m (0xFFFFFFFF: not finite)
__device__ __forceinline__ uint32_t coarse_margin_key(const float *__restrict__ yr, uint32_t dim, float cmax, float tau) {
    const uint32_t lane = threadIdx.x & 63;
    float yn = 0.0f;  // |y|^2 from an f32 sum (any order: its rounding is inside the factors below)
    for (uint32_t e = lane; e < dim; e += 64) yn = fmaf(yr[e], yr[e], yn);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) yn += __shfl_xor(yn, o, 64);
    const float rad = cmax + sqrtf(yn) * 1.000001f;
    const float mx = (0.00390625f + (float)(2 * dim + 64) * 5.9604645e-8f) * 1.05f * (rad * rad);
    float thr2 = tau + 2.0f * mx;
    thr2 = thr2 + fabsf(thr2) * 1.0e-6f;  // the comparison's own rounding
    const bool fin = fabsf(thr2) < 3.0e38f && fabsf(tau) < 3.0e38f;  // false for NaN / inf
    return fin ? ord32_biased(thr2) : 0xFFFFFFFFu;
}

template <int KPL>
__global__ __launch_bounds__(256) void select_refine_wave_kernel(float *__restrict__ dist, const float *__restrict__ y,
                                                                 const float *__restrict__ centroids, float cmax, uint32_t k, uint32_t dim,
                                                                 uint32_t nprobe, uint32_t *__restrict__ out_cluster,
                                                                 float *__restrict__ out_dist, uint32_t out_stride, uint32_t nq,
                                                                 unsigned long long *__restrict__ fallback_rows /* counter, may be null */) {
    __shared__ unsigned long long win[4][RQ_COARSE_CAND];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + wave;
    if (b >= nq) return;
    unsigned long long *wn = win[wave];
    float *d = dist + (uint64_t)b * k;
    const float *yr = y + (uint64_t)b * dim;
    const uint32_t grp = lane >> 3, al = lane & 7;  // 8 lanes per list: AVX lane al of list slot grp
    uint32_t key[KPL];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    auto list_of = [&](int i) { return 256u * (uint32_t)(i >> 2) + 4u * lane + (uint32_t)(i & 3); };
    const bool vec4 = (k & 3u) == 0u;
#pragma unroll
    for (int i4 = 0; i4 < KPL; i4 += 4) {
        const uint32_t j0 = list_of(i4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec4 && j0 < k) {
            v = *reinterpret_cast<const float4 *>(d + j0);
        } else if (!vec4) {
            if (j0 < k) v.x = d[j0];
            if (j0 + 1 < k) v.y = d[j0 + 1];
            if (j0 + 2 < k) v.z = d[j0 + 2];
            if (j0 + 3 < k) v.w = d[j0 + 3];
        }
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i4 + e;
            key[i] = 0xFFFFFFFFu;
            if (j0 + e < k) {
                key[i] = ord32_biased(ve[e]);
                kmin = key[i] < kmin ? key[i] : kmin;
                kmax = key[i] > kmax ? key[i] : kmax;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t a = __shfl_xor(kmin, o, 64), c = __shfl_xor(kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = c > kmax ? c : kmax;
    }
    auto count_le = [&](uint32_t t) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) c += (uint32_t)__popcll(__ballot(key[i] <= t));
        return c;
    };
    kmin = __builtin_amdgcn_readfirstlane(kmin), kmax = __builtin_amdgcn_readfirstlane(kmax);
    // ANY T with count(key <= T) >= nprobe bounds the nprobe-th smallest a' from above, which is all the candidate rule needs: the
    // bisection stops as soon as the count lands in [nprobe, nprobe + 12] (7-9 steps instead of the ~25 an exact threshold takes;
    // the price is up to 12 more candidates)
    uint32_t lo = kmin, hi = kmax;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t c = count_le(mid);
        if (c >= nprobe) {
            hi = mid;
            if (c <= nprobe + 12) break;
        } else {
            lo = mid + 1;
        }
    }
    const float tau = ord32_unbias(hi);
    const uint32_t T2 = coarse_margin_key(yr, dim, cmax, tau);
    const bool fin = T2 != 0xFFFFFFFFu;
    const uint32_t c2 = fin ? count_le(T2) : 0xFFFFFFFFu;
    if (!(c2 <= RQ_COARSE_CAND) || c2 < nprobe) {  // (wave-uniform) the plain way: every distance in exact order, then the exact selection
        for (uint32_t j0 = 0; j0 < k; j0 += 32) {
            uint32_t jj[4];
            float ee[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) jj[q] = j0 + 8 * q + grp < k ? j0 + 8 * q + grp : 0u;
            coarse_exact_dist4(centroids, yr, dim, al, jj, ee);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + 8 * q + grp < k && al == 0) d[j0 + 8 * q + grp] = ee[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own stores: read back below by other lanes of the wave
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (fallback_rows && lane == 0) atomicAdd(fallback_rows, 1ull);
        select_probe_wave<KPL>(dist, k, nprobe, out_cluster, out_dist, 0u, out_stride, b, wn);
        return;
    }
    // candidate ids into LDS (low half of the slots), in register order
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const bool take = key[i] <= T2;
        const uint64_t m = __ballot(take);
        if (m) {  // wave-uniform
            if (take) wn[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = list_of(i);
            base += (uint32_t)__popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    coarse_refine_tail(wn, c2, centroids, yr, dim, nprobe, b, out_cluster, out_dist, out_stride);
}


// The same for MORE lists than a wave can hold in registers (k > 8192: the probe ranking of a multi-GPU deployment is over the
// lists of ALL shards -- 32 768 at eight GPUs).  One sweep over the row of approximate distances leaves the minimum of every
// TILE of 32 consecutive lists (k / 32 keys: 16 per lane at k = 32 768).  The nprobe-th smallest tile minimum bounds the row's
// nprobe-th smallest a' from above (those nprobe minima are nprobe different lists), and tightly: the nearest lists of a query
// rarely share a tile.  Only the tiles whose minimum is within the margin are read again for the candidates (a few dozen
// 128-byte pieces instead of the row).  A row that cannot be handled (more than RQ_COARSE_CAND candidates, a margin that is not
// finite) gets all its distances in exact order and its flag set: select_probe_kernel then selects those rows (redo_flag).
// dynamic LDS: 4 x 64 TPL dwords (tile keys, then the flagged tiles, per wave)
template <int TPL>
__global__ __launch_bounds__(256) void select_refine_tiled_kernel(float *__restrict__ dist, const float *__restrict__ y,
                                                                  const float *__restrict__ centroids, float cmax, uint32_t k, uint32_t dim,
                                                                  uint32_t nprobe, uint32_t *__restrict__ out_cluster,
                                                                  float *__restrict__ out_dist, uint32_t out_stride, uint32_t nq,
                                                                  uint32_t *__restrict__ redo_flag,
                                                                  unsigned long long *__restrict__ fallback_rows /* counter, may be null */) {
    __shared__ unsigned long long win[4][RQ_COARSE_CAND];
    extern __shared__ __attribute__((aligned(16))) uint32_t tile_lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b = blockIdx.x * 4 + wave;
    if (b >= nq) return;
    unsigned long long *wn = win[wave];
    uint32_t *tkeys = tile_lds + wave * (64 * TPL);
    float *d = dist + (uint64_t)b * k;
    const float *yr = y + (uint64_t)b * dim;
    const uint32_t grp = lane >> 3, al = lane & 7;
    const uint32_t ntile = (k + 31) / 32;  // <= 64 TPL (host)
    const bool vec4 = (k & 3u) == 0u;
    // sweep: 32 tiles (1024 lists) per step -- four 16-byte loads in flight per lane --, 4 lists per lane and load, the 8 lanes of
    // a group fold one tile
    for (uint32_t t0 = 0; t0 < ntile; t0 += 32) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j0 = (t0 + 8 * u) * 32 + lane * 4;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vec4 && j0 < k) {
                v[u] = *reinterpret_cast<const float4 *>(d + j0);
            } else if (!vec4) {
                if (j0 < k) v[u].x = d[j0];
                if (j0 + 1 < k) v[u].y = d[j0 + 1];
                if (j0 + 2 < k) v[u].z = d[j0 + 2];
                if (j0 + 3 < k) v[u].w = d[j0 + 3];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j0 = (t0 + 8 * u) * 32 + lane * 4;
            const float ve[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t key = j0 + e < k ? ord32_biased(ve[e]) : 0xFFFFFFFFu;
                mn = key < mn ? key : mn;
            }
#pragma unroll
            for (int o = 4; o >= 1; o >>= 1) {
                const uint32_t a = __shfl_xor(mn, o, 8);
                mn = a < mn ? a : mn;
            }
            if (al == 0 && t0 + 8 * u + grp < 64 * TPL) tkeys[t0 + 8 * u + grp] = mn;  // (tiles past the last one come out as "no list")
        }
    }
    for (uint32_t i = ((ntile + 31) & ~31u) + lane; i < 64 * TPL; i += 64) tkeys[i] = 0xFFFFFFFFu;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    uint32_t tk[TPL];  // tile lane + 64 i
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int i = 0; i < TPL; ++i) {
        tk[i] = tkeys[lane + 64 * i];
        if (lane + 64 * i < ntile) {
            kmin = tk[i] < kmin ? tk[i] : kmin;
            kmax = tk[i] > kmax ? tk[i] : kmax;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t a = __shfl_xor(kmin, o, 64), c = __shfl_xor(kmax, o, 64);
        kmin = a < kmin ? a : kmin;
        kmax = c > kmax ? c : kmax;
    }
    kmin = __builtin_amdgcn_readfirstlane(kmin), kmax = __builtin_amdgcn_readfirstlane(kmax);
    auto count_le = [&](uint32_t t) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < TPL; ++i) c += (uint32_t)__popcll(__ballot(tk[i] <= t && lane + 64 * i < ntile));
        return c;
    };
    uint32_t lo = kmin, hi = kmax;  // (ntile >= nprobe: count_le(kmax) = ntile >= nprobe)
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t c = count_le(mid);
        if (c >= nprobe) {
            hi = mid;
            if (c <= nprobe + 12) break;
        } else {
            lo = mid + 1;
        }
    }
    // every lane has its tile keys in registers by now: the LDS copy becomes the list of flagged tiles
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // the lists with a' <= T (key << 32 | list id into the candidate slots): only tiles whose minimum is <= T are read again.
    // false: more than RQ_COARSE_CAND of them
    uint32_t base = 0;
    auto collect = [&](uint32_t T) -> bool {
        uint32_t nf = 0;
#pragma unroll
        for (int i = 0; i < TPL; ++i) {
            const bool take = tk[i] <= T && lane + 64 * i < ntile;
            const uint64_t m = __ballot(take);
            if (m) {
                if (take) tkeys[nf + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = lane + 64 * i;
                nf += (uint32_t)__popcll(m);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        base = 0;
        bool fits = true;
        constexpr int CU_ = RQ_COLLECT_UNROLL;
        for (uint32_t s0 = 0; fits && s0 < nf; s0 += 2 * CU_) {  // 2 CU_ flagged tiles per step (one per half-wave, CU_ loads in flight per lane:
                                                                 // one at a time the loop was a chain of ~50 dependent L2 round trips per query)
            uint32_t jv[CU_];
            float dv[CU_];
#pragma unroll
            for (int u = 0; u < CU_; ++u) {
                const uint32_t ti = s0 + 2 * u + (lane >> 5);
                jv[u] = ti < nf ? tkeys[ti] * 32 + (lane & 31) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < CU_; ++u) dv[u] = jv[u] < k ? d[jv[u]] : 0.0f;
#pragma unroll
            for (int u = 0; u < CU_; ++u) {
                const uint32_t key = ord32_biased(dv[u]);
                const bool take = jv[u] < k && key <= T;
                const uint64_t m = __ballot(take);
                if (m) {  // wave-uniform
                    const uint32_t cnt = (uint32_t)__popcll(m);
                    if (base + cnt > RQ_COARSE_CAND) fits = false;
                    else if (take) wn[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)key << 32) | jv[u];
                    base += cnt;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        return fits;
    };
    // first with the bound the tile minima give (tight when the nearest lists sit in different tiles)
    uint32_t T2 = ntile >= nprobe ? coarse_margin_key(yr, dim, cmax, ord32_unbias(hi)) : 0xFFFFFFFFu;
    bool ok = T2 != 0xFFFFFFFFu;
    if (ok && !collect(T2)) {
        // too many lists within the margin of that bound (wide margins: high dimensions): the row's nprobe-th smallest a' itself --
        // every list at or below the tile bound `hi` is collected (there are at least nprobe), bisection over those keys as the
        // single-wave kernel does over the row -- and the margin from there
        ok = collect(hi);
        if (ok) {
            uint32_t kk[RQ_COARSE_CAND / 64];
#pragma unroll
            for (int i = 0; i < (int)(RQ_COARSE_CAND / 64); ++i) kk[i] = lane + 64 * i < base ? (uint32_t)(wn[lane + 64 * i] >> 32) : 0xFFFFFFFFu;
            uint32_t l2 = kmin, h2 = hi;
            while (l2 < h2) {
                const uint32_t mid = l2 + ((h2 - l2) >> 1);
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < (int)(RQ_COARSE_CAND / 64); ++i) c += (uint32_t)__popcll(__ballot(kk[i] <= mid));
                if (c >= nprobe) {
                    h2 = mid;
                    if (c <= nprobe + 12) break;
                } else {
                    l2 = mid + 1;
                }
            }
            T2 = coarse_margin_key(yr, dim, cmax, ord32_unbias(h2));
            ok = T2 != 0xFFFFFFFFu && collect(T2);
        }
    }
    if (!ok || base < nprobe) {  // (wave-uniform) the plain way: every distance in exact order; the block-per-query selection takes the row
        for (uint32_t j0 = 0; j0 < k; j0 += 32) {
            uint32_t jj[4];
            float ee[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) jj[q] = j0 + 8 * q + grp < k ? j0 + 8 * q + grp : 0u;
            coarse_exact_dist4(centroids, yr, dim, al, jj, ee);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + 8 * q + grp < k && al == 0) d[j0 + 8 * q + grp] = ee[q];
        }
        if (lane == 0) {
            redo_flag[b] = 1u;
            if (fallback_rows) atomicAdd(fallback_rows, 1ull);
        }
        return;
    }
    if (lane == 0) redo_flag[b] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    coarse_refine_tail(wn, base, centroids, yr, dim, nprobe, b, out_cluster, out_dist, out_stride);
}

// ------------------------------------------------------------------------------------------------
// Shard merge (multi-GPU fan-out): per query, the m_out smallest of the `world` x `width` u64 keys the ranks
// contributed (all-gathered as in[world][nq][width], rows in any order), ascending.  Used for the probe lists
// (key = distance bits << 32 | list id) and for the per-shard top-k (key = Ord32 image << 32 | global id).
// One block per query, LDS bitonic sort; dynamic LDS: pow2_ceil(world * width) * 8 bytes.
// ------------------------------------------------------------------------------------------------
// rank_stride: keys between the blocks of consecutive ranks (nq * width, or more when a rank's block carries trailing words).
__global__ __launch_bounds__(256) void merge_smallest_u64_kernel(const unsigned long long *__restrict__ in, uint32_t world,
                                                                 uint32_t nq, uint32_t width, uint32_t m_out,
                                                                 unsigned long long *__restrict__ out, uint64_t rank_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char merge_smem[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(merge_smem);
    const uint32_t b = blockIdx.x, m = world * width;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const uint32_t w = i / width, e = i - w * width;
        keys[i] = in[(uint64_t)w * rank_stride + (uint64_t)b * width + e];
    }
    __syncthreads();
    bitonic_sort_block(keys, m, [](unsigned long long v) { return v; });
    for (uint32_t i = threadIdx.x; i < m_out; i += blockDim.x) out[(uint64_t)b * m_out + i] = i < m ? keys[i] : ~0ull;
}

// ------------------------------------------------------------------------------------------------
// Per-(query, probed list) query quantisation (src/rabitq.rs:304-317):
//   residual = y - c (src/simd.rs:138), (lo, hi) (:143-157), delta = (hi - lo) * (1/15),
//   q = cvtps_epi32((res - lo) * (1/delta)) (:215, sub then mul, RNE, no bias), sum of q,
//   4 bit planes, bit b of word w <-> dimension 64w + b (:103).
// One wave per pair; lane holds dimensions {lane, 64+lane, ...}; a plane word is one __ballot.
// `pair_cluster[p]` is the list paired with rotated query row `pair_row[p]` (or p / nprobe).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_kernel(const float *__restrict__ y,
                                                   const float *__restrict__ centroids,
                                                   const uint32_t *__restrict__ offsets,
                                                   const uint32_t *__restrict__ pair_cluster,
                                                   const float *__restrict__ pair_ycd, uint32_t npairs,
                                                   uint32_t pairs_per_row, uint32_t dim,
                                                   PairScalars *__restrict__ scal,
                                                   uint64_t *__restrict__ planes,
                                                   uint32_t *__restrict__ qnib,
                                                   uint32_t *__restrict__ qf6,
                                                   uint32_t *__restrict__ out_sum_u32, uint32_t nlists,
                                                   uint32_t skip_empty) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npairs) return;
    const uint32_t row = p / pairs_per_row;
    const uint32_t c = pair_cluster[p];
    const uint32_t len_c = c < nlists ? offsets[c + 1] - offsets[c] : 0u;
    if (len_c == 0 && skip_empty) {  // nothing to scan for this pair (e.g. a list another shard owns)
        if (lane == 0) {
            PairScalars s;
            s.lower = 0.0f, s.delta = 0.0f, s.sumq = 0.0f, s.ycd = pair_ycd[p], s.ycd_sqrt = 0.0f;
            s.row = row, s.list_begin = 0, s.list_len = 0, s.stream_begin = 0, s.pad = 0;
            scal[p] = s;
            if (out_sum_u32) out_sum_u32[p] = 0;
        }
        return;
    }
    const float *yr = y + (uint64_t)row * dim;
    const float *cr = centroids + (uint64_t)c * dim;
    const uint32_t W = dim >> 6;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    for (uint32_t w = 0; w < W; ++w) {
        float r = yr[64 * w + lane] - cr[64 * w + lane];
        mn = r < mn ? r : mn;
        mx = r > mx ? r : mx;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        float a = __shfl_xor(mn, o, 64), bb = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn;
        mx = bb > mx ? bb : mx;
    }
    const float scalar = 1.0f / 15.0f;          // consts.rs:10
    const float delta = (mx - mn) * scalar;     // rabitq.rs:307
    const float one_over_delta = 1.0f / delta;  // :308 f32::recip
    uint32_t sum = 0;
    uint64_t *pl = planes + (uint64_t)p * 4 * W;
    for (uint32_t w = 0; w < W; ++w) {
        float r = yr[64 * w + lane] - cr[64 * w + lane];
        int32_t q = cvtps_epi32((r - mn) * one_over_delta);
        sum += (uint32_t)q;
        if (planes) {
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {
                uint64_t word = __ballot((q >> bit) & 1);
                if (lane == 0) pl[bit * W + w] = word;
            }
        }
        if (qf6) {  // matrix-core operand: q/2 as fp6 e2m3 (every integer 0..15 is exact), 6-bit fields packed as a
                    // little-endian bit stream; lane half h = lane>>5 of word w <-> the 6 dwords [h][w][0..6)
            const uint32_t v = (uint32_t)q & 15u;
            const uint32_t c6 = v < 4 ? 4 * v : (v < 8 ? 8 + 2 * v : 16 + v);
            const uint32_t i16 = lane & 15, bitpos = 6 * i16, d0 = bitpos >> 5, off = bitpos & 31;
            const uint64_t wide = (uint64_t)c6 << off;
            const uint32_t lo32 = (uint32_t)wide, hi32 = (uint32_t)(wide >> 32);
            uint32_t w0 = d0 == 0 ? lo32 : 0u;
            uint32_t w1 = d0 == 0 ? hi32 : (d0 == 1 ? lo32 : 0u);
            uint32_t w2 = d0 == 1 ? hi32 : (d0 == 2 ? lo32 : 0u);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                w0 |= __shfl_xor(w0, o, 16);
                w1 |= __shfl_xor(w1, o, 16);
                w2 |= __shfl_xor(w2, o, 16);
            }
            if (i16 < 3)
                qf6[(uint64_t)p * 12 * W + (lane >> 5) * 6 * W + 6 * w + 3 * ((lane >> 4) & 1) + i16] =
                    i16 == 0 ? w0 : (i16 == 1 ? w1 : w2);
        }
        if (qnib) {  // the same 4-bit codes packed 8 per dword (dword m <-> dims 8m..8m+7, nibble i <-> dim 8m+i):
                     // the operand form of v_dot8_u32_u4 used by the fused scan kernel
            uint32_t nib = ((uint32_t)q & 15u) << (4 * (lane & 7));
            nib |= __shfl_xor(nib, 1, 8);
            nib |= __shfl_xor(nib, 2, 8);
            nib |= __shfl_xor(nib, 4, 8);
            if ((lane & 7) == 0) qnib[(uint64_t)p * 8 * W + 8 * w + (lane >> 3)] = nib;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) {
        PairScalars s;
        float ycd = pair_ycd[p];
        s.lower = mn;
        s.delta = delta;
        s.sumq = (float)sum;  // rabitq.rs:322 `scalar_sum as f32`
        s.ycd = ycd;
        s.ycd_sqrt = sqrtf(ycd);  // :346
        s.row = row;
        s.list_begin = offsets[c];
        s.list_len = offsets[c + 1] - offsets[c];
        s.stream_begin = 0;  // filled by pair_prefix_kernel
        s.pad = 0;
        scal[p] = s;
        if (out_sum_u32) out_sum_u32[p] = sum;
    }
}

// The same quantisation for dim in {64, 128, 256, 512, 768, 1024} with LP lanes per pair (several pairs per wave
// up to dim 128, R rounds of 256 dimensions from dim 512): every lane
// owns 4 consecutive dimensions (one 16-byte load of y and of the centroid), the reductions take log2(LP)
// shuffle steps, and the operand images come out of one neighbour exchange each: a lane's four 4-bit codes
// are half a qnib dword, its four fp6 fields 24 bits of the 6-dword image of its 32-dimension block.
// Writes the fused scan's operands only (no bit planes); arithmetic identical to prep_kernel.
// The pairs p0, p0 + 64/LP, ... (PP of them) of one lane group; p0 already includes the group's index lane / LP.
template <int LP, int R, int PP>
__device__ __forceinline__ void prep_small_pairs(const float *__restrict__ y, const float *__restrict__ centroids,
                                                 const uint32_t *__restrict__ offsets,
                                                 const uint32_t *__restrict__ pair_cluster,
                                                 const float *__restrict__ pair_ycd, uint32_t npairs,
                                                 uint32_t pairs_per_row, PairScalars *__restrict__ scal,
                                                 uint32_t *__restrict__ qnib, uint32_t *__restrict__ qf6,
                                                 uint32_t nlists, uint32_t skip_empty, uint32_t p0,
                                                 const uint32_t *__restrict__ live_list = nullptr /* compacted pair ids (pair_split_kernel): p0 indexes it */,
                                                 uint32_t qn_slots = 0xFFFFFFFFu /* the 4-bit operand is written for probe slots below this only */) {
    // dim = 4 * LP * R: LP lanes per pair, each owning 4 consecutive dimensions in each of R rounds of 4*LP
    // dimensions (R > 1 only with LP = 64: dim 512, 768, 1024).  Every lane group handles PP pairs: the kernel is a
    // chain of dependent gathers (probe list -> centroid row, list bounds), so the loads of all PP pairs are issued
    // before any of them is consumed.
    static_assert(R >= 1 && (LP == 16 || LP == 32 || LP == 64), "lane groups of 16 / 32 / 64; a round = 4 LP dimensions (the packing's neighbour exchanges stay inside a round)");
    constexpr uint32_t DIM = 4 * LP * R, W = DIM / 64, PPW = 64 / LP;
    const uint32_t lane = threadIdx.x & 63, sub = lane % LP;
    uint32_t cl[PP], lb[PP], ll[PP];
    float ycd_in[PP];
    bool live[PP];
    uint32_t pid[PP];  // the pair each of the PP rounds works on
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const uint32_t pi = p0 + pp * PPW;
        live[pp] = pi < npairs;  // uniform over the pair's LP lanes
        pid[pp] = live_list ? (live[pp] ? live_list[pi] : 0u) : pi;
        const uint32_t p = pid[pp];
        cl[pp] = live[pp] ? pair_cluster[p] : 0xFFFFFFFFu;
        ycd_in[pp] = live[pp] ? pair_ycd[p] : 0.0f;
    }
    float4 cv[PP][R], yv[PP][R];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const uint32_t p = pid[pp], c = cl[pp];
        const bool in = live[pp] && c < nlists;
        lb[pp] = in ? offsets[c] : 0u;
        ll[pp] = in ? offsets[c + 1] - lb[pp] : 0u;
        const uint32_t row = live[pp] ? p / pairs_per_row : 0u;
        // skip_empty == 2 (an index with many empty lists: a shard of a multi-GPU deployment, where 7 of 8 probed lists live on
        // other ranks): the query and centroid rows of a pair are only fetched once its list is known to have members -- one more
        // dependent round trip for the pairs that stay, 1 KB less traffic for each that goes
        const bool fetch = skip_empty == 2 ? ll[pp] != 0 : true;
#pragma unroll
        for (int rd = 0; rd < R; ++rd) {
            if (fetch) {
                yv[pp][rd] = *reinterpret_cast<const float4 *>(y + (uint64_t)row * DIM + 4 * LP * rd + 4 * sub);
                cv[pp][rd] = *reinterpret_cast<const float4 *>(centroids + (uint64_t)(in ? c : 0u) * DIM + 4 * LP * rd + 4 * sub);
            } else {
                yv[pp][rd] = make_float4(0.0f, 0.0f, 0.0f, 0.0f), cv[pp][rd] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
    }
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const uint32_t p = pid[pp];
        if (!live[pp]) continue;
        const uint32_t row = p / pairs_per_row;
        if (ll[pp] == 0 && skip_empty) {  // nothing to scan for this pair (e.g. a list another shard owns)
            if (sub == 0) {
                PairScalars s;
                s.lower = 0.0f, s.delta = 0.0f, s.sumq = 0.0f, s.ycd = ycd_in[pp], s.ycd_sqrt = 0.0f;
                s.row = row, s.list_begin = 0, s.list_len = 0, s.stream_begin = 0, s.pad = 0;
                scal[p] = s;
            }
            continue;
        }
        float r[R][4];
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
#pragma unroll
        for (int rd = 0; rd < R; ++rd) {
            r[rd][0] = yv[pp][rd].x - cv[pp][rd].x, r[rd][1] = yv[pp][rd].y - cv[pp][rd].y;
            r[rd][2] = yv[pp][rd].z - cv[pp][rd].z, r[rd][3] = yv[pp][rd].w - cv[pp][rd].w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mn = r[rd][e] < mn ? r[rd][e] : mn;
                mx = r[rd][e] > mx ? r[rd][e] : mx;
            }
        }
#pragma unroll
        for (int o = LP / 2; o >= 1; o >>= 1) {
            const float a = __shfl_xor(mn, o, LP), bb = __shfl_xor(mx, o, LP);
            mn = a < mn ? a : mn;
            mx = bb > mx ? bb : mx;
        }
        const float scalar = 1.0f / 15.0f;          // consts.rs:10
        const float delta = (mx - mn) * scalar;     // rabitq.rs:307
        const float one_over_delta = 1.0f / delta;  // :308 f32::recip
        uint32_t sum = 0;
#pragma unroll
        for (int rd = 0; rd < R; ++rd) {
            uint32_t nib16 = 0, f24 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int32_t q = cvtps_epi32((r[rd][e] - mn) * one_over_delta);
                sum += (uint32_t)q;
                const uint32_t v = (uint32_t)q & 15u;
                nib16 |= v << (4 * e);
                f24 |= (v < 4 ? 4 * v : (v < 8 ? 8 + 2 * v : 16 + v)) << (6 * e);  // q/2 as fp6 e2m3
            }
            const uint32_t dsub = LP * rd + sub;  // this lane's 4-dimension group among the dim/4 of the vector
            {  // qnib: dword m <-> dims 8m..8m+7 = groups 2m (low half), 2m+1 (high half)
                const uint32_t other = __shfl_xor(nib16, 1, LP);
                if (qnib && (sub & 1) == 0 && p - row * pairs_per_row < qn_slots) qnib[(uint64_t)p * 8 * W + (dsub >> 1)] = nib16 | (other << 16);
            }
            {  // qf6: group t = dsub % 8 of a 32-dimension block holds stream bits [24t, 24t+24) of its 6 dwords
                const uint32_t nxt = __shfl_down(f24, 1, LP);
                const uint32_t t = dsub & 7, sh = 8 * (t & 3);
                if (qf6 && (t & 3) != 3) {
                    const uint32_t w = dsub >> 4, h = (dsub >> 3) & 1;
                    qf6[(uint64_t)p * 12 * W + h * 6 * W + 6 * w + 3 * (t >> 2) + (t & 3)] = (f24 >> sh) | (nxt << (24 - sh));
                }
            }
        }
#pragma unroll
        for (int o = LP / 2; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, LP);
        if (sub == 0) {
            PairScalars s;
            s.lower = mn;
            s.delta = delta;
            s.sumq = (float)sum;  // rabitq.rs:322 `scalar_sum as f32`
            s.ycd = ycd_in[pp];
            s.ycd_sqrt = sqrtf(ycd_in[pp]);  // :346
            s.row = row;
            s.list_begin = lb[pp];
            s.list_len = ll[pp];
            s.stream_begin = 0;  // filled by pair_prefix_kernel
            s.pad = 0;
            scal[p] = s;
        }
    }
}

template <int LP, int R, int PP>
__global__ __launch_bounds__(256) void prep_small_kernel(const float *__restrict__ y,
                                                         const float *__restrict__ centroids,
                                                         const uint32_t *__restrict__ offsets,
                                                         const uint32_t *__restrict__ pair_cluster,
                                                         const float *__restrict__ pair_ycd, uint32_t npairs,
                                                         uint32_t pairs_per_row, PairScalars *__restrict__ scal,
                                                         uint32_t *__restrict__ qnib, uint32_t *__restrict__ qf6,
                                                         uint32_t nlists, uint32_t skip_empty, uint32_t qn_slots) {
    constexpr uint32_t PPW = 64 / LP;
    const uint32_t p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (PPW * PP) + (threadIdx.x & 63) / LP;
    prep_small_pairs<LP, R, PP>(y, centroids, offsets, pair_cluster, pair_ycd, npairs, pairs_per_row, scal, qnib, qf6, nlists,
                                skip_empty, p0, nullptr, qn_slots);
}

// Sharded passes (a rank of a multi-GPU deployment: most probed lists live on other ranks, i.e. are empty here): one THREAD
// per pair writes the scalars of the pairs with nothing to scan and lists the others, and the quantisation kernel -- a lane
// group per pair -- then runs over that list only (7 of 8 lane groups did nothing but find their list empty).
__global__ __launch_bounds__(1024) void pair_split_kernel(const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ pair_cluster,
                                                          const float *__restrict__ pair_ycd, uint32_t npairs, uint32_t pairs_per_row,
                                                          uint32_t nlists, PairScalars *__restrict__ scal,
                                                          uint32_t *__restrict__ live_list, uint32_t *__restrict__ live_count) {
    // 4096 pairs per block, ONE reservation per block on the list's counter (a wave-level reservation each was half a million
    // atomics on one address per pass: 3 ms)
    __shared__ uint32_t wcnt[4][16], s_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool keep[4];
    uint32_t rank[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t p = blockIdx.x * 4096 + u * 1024 + threadIdx.x;
        keep[u] = false;
        if (p < npairs) {
            const uint32_t c = pair_cluster[p];
            const uint32_t len = c < nlists ? offsets[c + 1] - offsets[c] : 0u;
            keep[u] = len != 0;
            if (!keep[u]) {  // exactly what prep_small_pairs writes for such a pair
                PairScalars s;
                s.lower = 0.0f, s.delta = 0.0f, s.sumq = 0.0f, s.ycd = pair_ycd[p], s.ycd_sqrt = 0.0f;
                s.row = p / pairs_per_row, s.list_begin = 0, s.list_len = 0, s.stream_begin = 0, s.pad = 0;
                scal[p] = s;
            }
        }
        const uint64_t m = __ballot(keep[u]);
        rank[u] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[u][wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int u = 0; u < 4; ++u)
            for (int w = 0; w < 16; ++w) {
                const uint32_t c = wcnt[u][w];
                wcnt[u][w] = tot;
                tot += c;
            }
        s_base = tot ? atomicAdd(live_count, tot) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (keep[u]) live_list[s_base + wcnt[u][wave] + rank[u]] = blockIdx.x * 4096 + u * 1024 + threadIdx.x;
}
template <int LP, int R, int PP>
__global__ __launch_bounds__(256) void prep_small_listed_kernel(const float *__restrict__ y,
                                                                const float *__restrict__ centroids,
                                                                const uint32_t *__restrict__ offsets,
                                                                const uint32_t *__restrict__ pair_cluster,
                                                                const float *__restrict__ pair_ycd, const uint32_t *__restrict__ live_list,
                                                                uint32_t nlive,
                                                                uint32_t pairs_per_row, PairScalars *__restrict__ scal,
                                                                uint32_t *__restrict__ qnib, uint32_t *__restrict__ qf6,
                                                                uint32_t nlists) {
    constexpr uint32_t PPW = 64 / LP;
    const uint32_t p0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (PPW * PP) + (threadIdx.x & 63) / LP;
    prep_small_pairs<LP, R, PP>(y, centroids, offsets, pair_cluster, pair_ycd, nlive, pairs_per_row, scal, qnib, qf6, nlists, 1u, p0, live_list);
}

// Position of every probed list in the query's candidate stream (the order the reference visits
// candidates: lists nearest-first, members in stored order) and the stream length, which is also
// what the reference adds to METRICS.rough for this query (src/rerank.rs:105).
// one wave, query row b: 64 slots per step, exclusive scan by shuffles, carry across steps
__device__ __forceinline__ void pair_prefix_row(PairScalars *__restrict__ scal, uint32_t b, uint32_t nprobe,
                                                unsigned long long *__restrict__ rough_count) {
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long carry = 0;
    for (uint32_t s0 = 0; s0 < nprobe; s0 += 64) {
        const uint32_t s = s0 + lane;
        PairScalars *ps = scal + (uint64_t)b * nprobe + s;
        const unsigned long long len = s < nprobe ? ps->list_len : 0;
        unsigned long long incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        const unsigned long long begin = carry + incl - len;
        if (s < nprobe) ps->stream_begin = begin > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)begin;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) rough_count[b] = carry;
}
__global__ __launch_bounds__(256) void pair_prefix_kernel(PairScalars *__restrict__ scal, uint32_t nq, uint32_t nprobe,
                                                          unsigned long long *__restrict__ rough_count) {
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b < nq) pair_prefix_row(scal, b, nprobe, rough_count);
}

// ------------------------------------------------------------------------------------------------
// Work-list construction for a stage: the (query, slot) pairs with slot in [slot_lo, slot_hi).
// pair-major: one group per pair.  cluster-major: pairs bucketed by list so that a list is read
// from HBM once and scored against every query probing it.
// ------------------------------------------------------------------------------------------------
// does list [begin, begin+len) of the candidate stream intersect the stage [s_lo, s_hi) ?
__device__ __forceinline__ bool pair_in_stage(const PairScalars &ps, uint32_t s_lo, uint32_t s_hi) {
    const uint64_t b = ps.stream_begin, e = b + ps.list_len;
    return ps.list_len != 0 && b < s_hi && e > s_lo;
}

// `count` = nq * slot_hi work items: only the first slot_hi slots of every query can be in the stage
__global__ void group_count_kernel(const PairScalars *__restrict__ scal,
                                   const uint32_t *__restrict__ probe_cluster, uint32_t count, uint32_t nprobe,
                                   uint32_t slot_hi, uint32_t s_lo, uint32_t s_hi, uint32_t *__restrict__ grp_cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t b = i / slot_hi, p = b * nprobe + (i - b * slot_hi);
    if (pair_in_stage(scal[p], s_lo, s_hi)) atomicAdd(&grp_cnt[probe_cluster[p]], 1u);
}

// The same count for big stages, with the place of every pair inside its list's group decided on the way
// (cluster-major stages of large batches: ~16 pairs per list and block).  A block owns RQ_RANK_ITEMS consecutive work
// items and a histogram of the lists in LDS: an item's rank inside (block, list) comes from the LDS atomic, the
// block's base inside the list from ONE global atomic per (block, list) -- instead of one contended global atomic
// per pair here and a second one in stage_fill_kernel (4.2 M each at batch 65 536, on 4096 addresses).
//   rank[i]           place of item i among its block's pairs of the same list (~0u: not in the stage)
//   blk_base[blk][c]  first place of block blk's pairs in list c's group
// Dynamic LDS: k counters.
#define RQ_RANK_ITEMS 32768u
__global__ __launch_bounds__(1024) void group_rank_kernel(const PairScalars *__restrict__ scal,
                                                          const uint32_t *__restrict__ probe_cluster, uint32_t count,
                                                          uint32_t nprobe, uint32_t slot_hi, uint32_t s_lo, uint32_t s_hi,
                                                          uint32_t k, uint32_t *__restrict__ grp_cnt,
                                                          uint32_t *__restrict__ rank, uint32_t *__restrict__ blk_base) {
    extern __shared__ uint32_t rank_hist[];
    for (uint32_t c = threadIdx.x; c < k; c += 1024) rank_hist[c] = 0;
    __syncthreads();
    const uint32_t first = blockIdx.x * RQ_RANK_ITEMS;
    for (uint32_t j = threadIdx.x; j < RQ_RANK_ITEMS; j += 1024) {
        const uint32_t i = first + j;
        if (i >= count) break;
        const uint32_t b = i / slot_hi, p = b * nprobe + (i - b * slot_hi);
        uint32_t r = ~0u;
        if (pair_in_stage(scal[p], s_lo, s_hi)) r = atomicAdd(&rank_hist[probe_cluster[p]], 1u);
        rank[i] = r;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < k; c += 1024) {
        const uint32_t h = rank_hist[c];
        blk_base[(uint64_t)blockIdx.x * k + c] = h ? atomicAdd(&grp_cnt[c], h) : 0u;
    }
}

// exclusive scan of cnt[0..k) into start[0..k]; single block, any k.  Also zeroes cnt for the
// fill pass (cnt is reused as the per-list cursor, so it holds the counts again afterwards).
// pad32 bit 0: every group starts at a multiple of 32 records (the matrix-core scan's query tiles).
// pad32 (matrix-core stages): the rows between a group's last record and its 32-row boundary are marked "no query"
// right here (their threshold operand: constant term -inf, so the accumulator of such a row starts at -inf and is
// never flagged): the scan then needs no per-tile masking.  recs / opdw: the stage's tile images.
__global__ __launch_bounds__(1024) void group_scan_kernel(uint32_t *__restrict__ cnt, uint32_t k,
                                                          uint32_t *__restrict__ start, uint32_t pad32,
                                                          uint32_t *__restrict__ recs, uint32_t opdw) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < k; base += 1024) {
        uint32_t i = base + tid;
        uint32_t v = i < k ? cnt[i] : 0;
        if (pad32 & 1u) v = (v + 31u) & ~31u;
        uint32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t up = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (uint32_t w = 0; w < wid; ++w) woff += wsum[w];
        uint32_t c0 = carry;
        if (i < k) {
            const uint32_t st0 = c0 + woff + incl - v;
            start[i] = st0;
            if ((pad32 & 1u) && recs) {
                const uint32_t real = cnt[i];
                const bool additive = (pad32 & 4u) != 0;  // bit 2: tile images of the additive gate (start value -inf)
                const uint32_t opld = rq_img_opld(opdw, additive), img = rq_img_dwords(opdw, additive);
                for (uint32_t at = st0 + real; at < st0 + v; ++at) {  // at most 31 rows
                    if (additive) {
                        recs[(uint64_t)(at >> 5) * img + 32 * opld + 32 * RQ_RECA_TAIL + (at & 31u)] = 0xFF800000u;
                        continue;
                    }
                    uint32_t *tl = recs + (uint64_t)(at >> 5) * img + 32 * opld + (at & 31u) * RQ_REC_TAIL + RQ_REC_V0;
                    *reinterpret_cast<uint4 *>(tl) = make_uint4(0u, 0u, 0u, 0x0000FF80u);  // slots 0..7: c0 = -inf
                    *reinterpret_cast<uint4 *>(tl + 4) = make_uint4(0u, 0u, 0u, 0u);        // slots 8..15
                }
            }
            if (!(pad32 & 2u)) cnt[i] = 0;  // bit 1: the places are already known (group_rank_kernel), cnt stays the count
        }
        __syncthreads();
        if (tid == 1023) carry = c0 + woff + incl;
        __syncthreads();
    }
    if (tid == 0) start[k] = carry;
}

// f32 -> bf16 bits (round to nearest even) and back; finite inputs well inside the f32 range
__device__ __forceinline__ uint32_t bf16_rne(float x) {
    uint32_t u = __builtin_bit_cast(uint32_t, x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }

// work item wi = (query, slot < slot_hi), handled by the 16 lanes threadIdx.x & ~15 .. | 15; thr_b: the query's threshold
template <int LPP = 16 /* lanes per work item: 16, or 8 for the additive tile images (16-byte aligned operand rows) */>
__device__ __forceinline__ void stage_fill_item(const PairScalars *__restrict__ scal,
                                                const uint32_t *__restrict__ probe_cluster,
                                                const uint32_t *__restrict__ operand /* opdw dwords per pair */,
                                                const float *__restrict__ thr, uint32_t wi,
                                                uint32_t nprobe, uint32_t slot_hi, uint32_t opdw, uint32_t s_lo,
                                                uint32_t s_hi,
                                                uint32_t cluster_major,
                                                const uint32_t *__restrict__ grp_start,
                                                uint32_t *__restrict__ grp_cursor,
                                                uint32_t *__restrict__ recs, const FactorStats &fs,
                                                uint32_t tile_images /* 0 record-major, 1 tile images, 2 tile images of the additive gate */,
                                                const uint32_t *__restrict__ rank /* group_rank_kernel, or null */,
                                                const uint32_t *__restrict__ blk_base, uint32_t k,
                                                const float4 *__restrict__ list_uref /* tile_images == 2: U0 per list */) {
    const uint32_t sub = threadIdx.x & (LPP - 1);                // LPP lanes per pair
    // (ranked placement: the counting pass has already decided which items are in the stage -- 7 of 8 are not when the pairs of
    // seven other shards' lists ride along in a multi-GPU pass; they leave before their 40-byte scalars are fetched)
    if (cluster_major && rank && rank[wi] == ~0u) return;
    const uint32_t wb = wi / slot_hi, p = wb * nprobe + (wi - wb * slot_hi);
    const PairScalars ps = scal[p];
    const bool in = pair_in_stage(ps, s_lo, s_hi);
    if (cluster_major && !in) return;
    uint32_t at = p;
    uint32_t list_id = 0;
    if (cluster_major) {
        const uint32_t c = probe_cluster[p];
        list_id = c;
        if (rank) {  // places were handed out by group_rank_kernel
            at = grp_start[c] + blk_base[(uint64_t)(wi / RQ_RANK_ITEMS) * k + c] + rank[wi];
        } else {
            uint32_t a = 0;
            if (sub == 0) a = atomicAdd(&grp_cursor[c], 1u);
            at = grp_start[c] + __shfl(a, 0, LPP);
        }
    }
    // record-major: record `at` = opdw operand dwords + RQ_REC_TAIL tail dwords.  tile images (matrix-core
    // scan): records in tiles of 32, each tile stored as the exact LDS image the kernel copies in with
    // LDS-DMA: 32 operand rows of opdw+2 dwords, then the 32 tails.
    const uint32_t stride = opdw + RQ_REC_TAIL;
    uint32_t *r = recs + (uint64_t)at * stride;
    uint32_t *tdst = r + opdw;
    float *cdst = nullptr;  // additive gate: the row's accumulator start value
    if (tile_images) {
        const uint32_t opld = rq_img_opld(opdw, tile_images == 2), img = rq_img_dwords(opdw, tile_images == 2);
        const uint32_t taild = tile_images == 2 ? RQ_RECA_TAIL : RQ_REC_TAIL;
        uint32_t *base = recs + (uint64_t)(at >> 5) * img;
        r = base + (at & 31u) * opld;
        tdst = base + 32 * opld + (at & 31u) * taild;
        if (tile_images == 2) cdst = reinterpret_cast<float *>(base + 32 * opld + 32 * taild + (at & 31u));
    }
    if constexpr (LPP == 8) {  // record-major records (stride opdw + 20 dwords) and additive tile images (rows of opdw + 4 dwords): 16-byte aligned like the operand itself (opdw = 8 W or 12 W)
        const uint4 *src = reinterpret_cast<const uint4 *>(operand + (uint64_t)p * opdw);
        uint4 *dst = reinterpret_cast<uint4 *>(r);
        for (uint32_t i = sub; i < opdw / 4; i += 8) dst[i] = src[i];
    } else {  // operand rows are 8-byte aligned in both layouts (opdw, the record stride and the image row stride are even)
        const uint2 *src = reinterpret_cast<const uint2 *>(operand + (uint64_t)p * opdw);
        uint2 *dst = reinterpret_cast<uint2 *>(r);
        for (uint32_t i = sub; i < opdw / 2; i += 16) dst[i] = src[i];
    }
    {  // the tail: computed by every lane of the pair (they would idle otherwise), stored 16 bytes per lane by lanes 0..4
        uint32_t t[RQ_REC_TAIL];
        uint32_t lo = 0, hi = 0;
        if (in) {
            lo = s_lo > ps.stream_begin ? s_lo - ps.stream_begin : 0u;
            hi = s_hi - ps.stream_begin;  // in-stage => stream_begin < s_hi
            hi = hi < ps.list_len ? hi : ps.list_len;
        }
        t[RQ_REC_LOWER] = __builtin_bit_cast(uint32_t, ps.lower);
        t[RQ_REC_DELTA] = __builtin_bit_cast(uint32_t, ps.delta);
        t[RQ_REC_SUMQ] = __builtin_bit_cast(uint32_t, ps.sumq);
        t[RQ_REC_YCD] = __builtin_bit_cast(uint32_t, ps.ycd);
        t[RQ_REC_YCD_SQRT] = __builtin_bit_cast(uint32_t, ps.ycd_sqrt);
        t[RQ_REC_THR] = __builtin_bit_cast(uint32_t, thr[ps.row]);
        t[RQ_REC_LO] = lo;
        t[RQ_REC_HI] = hi;
        t[RQ_REC_ROW] = ps.row;
        t[RQ_REC_SLOT] = p - ps.row * nprobe;
        t[RQ_REC_LIST_BEGIN] = ps.list_begin;
        t[RQ_REC_LIST_LEN] = ps.list_len;
        // Integer form of the gate (used by the matrix-core scan).  With F = factor_ip * delta < 0,
        //   rough < thr  <=>  s > S* = [ (thr - ycd) + (-1) cds + (-lower) ppc + ysq eb ] / (2 F) + sumq / 2
        // (real arithmetic), a rank-5 bilinear form in u'_c = (1, cds, ppc, eb)/fip, 1  and v'_q.  The scan
        // starts its accumulator tile at -S*/2 (its dot products come out as s/2) with ONE
        // v_mfma_f32_32x32x16_bf16: every u', v' is split into bf16 hi + lo and the products
        // hi*hi + hi*lo + lo*hi are summed (12 slots), the constant term is split three ways (exact), 1
        // slot is unused; the gate is then "accumulator > 0".  S* is lowered by a margin that covers the
        // f32 rounding of the exact expression (2, safe while the terms stay below 2^19), the dropped
        // lo*lo products and split residues (< 2^-14 of the sum of |terms|) and the roundings of adding
        // s/2 onto -S*/2 inside the matrix unit (< 2^-19 of the magnitudes), so the test can never hide a
        // candidate the exact f32 expression would pass; where the scales make the bound unsafe (or
        // delta <= 0) the query is marked "always flagged" (-S* = +inf) and every candidate takes the
        // exact path.
        const float th = thr[ps.row];
        const float inv2d = 0.5f / ps.delta;
        float v[4] = {(th - ps.ycd) * inv2d, -inv2d, -ps.lower * inv2d, ps.ycd_sqrt * inv2d};
        const float qb = (fabsf(th - ps.ycd) + fs.cds_max + fabsf(ps.lower) * fs.ppc_absmax + ps.ycd_sqrt * fs.eb_max) *
                         fs.invfip_absmax * fabsf(inv2d);
        const bool safe = ps.delta > 0.0f && qb + ps.sumq < 524288.0f;  // also false for NaN / inf
        const float margin = 2.0f + qb * (1.0f / 8192.0f) + (qb + ps.sumq) * (1.0f / 262144.0f);
        if (tile_images == 2) {
            // Additive gate (scan_mfma_kernel<.., ADD>): the query's side of  S* >= B_q + G_c,  B_q = sum_r U0[r] v'_q[r] + sumq / 2 with
            // the list's reference U0 (U0[2] = 0), as the accumulator's start value C_q = -(B_q - margin) / 2 (the dot products come
            // out as s / 2): "s/2 + C_q > G_c / 2" then holds for every candidate the exact f32 expression would pass.  The margin is
            // the bf16 form's (2 for the f32 rounding of the exact expression; qb 2^-13 covers, many times over, the f32 roundings
            // of B_q, of v' inside the list's V0 / DV and of the candidate's u' -- all relative 2^-22 of terms bounded by qb) with a
            // wider share for the matrix unit's own accumulation of C_q + s/2 (f32, magnitudes below qb + sumq).
            const float4 u0 = list_uref[list_id];
            float bq = u0.x * v[0];
            bq += u0.y * v[1];
            bq += u0.w * v[3];
            bq += 0.5f * ps.sumq;
            const float margin_a = 2.0f + qb * (1.0f / 8192.0f) + (qb + ps.sumq) * (1.0f / 65536.0f);
            float cq = -0.5f * (bq - margin_a);
            if (!safe || !(fabsf(cq) < 1.0e37f)) cq = __builtin_inff();  // always flagged: the exact path decides
            if (sub == 5) *cdst = cq;
            if (sub < 3) *reinterpret_cast<uint4 *>(tdst + 4 * sub) = make_uint4(t[4 * sub], t[4 * sub + 1], t[4 * sub + 2], t[4 * sub + 3]);
            return;
        }
        const float v4 = -0.5f * (0.5f * ps.sumq - margin);
        uint32_t vh[4], vl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = -0.5f * v[i];
            vh[i] = bf16_rne(x);
            vl[i] = bf16_rne(x - bf16_to_f32(vh[i]));
        }
        uint32_t c0 = bf16_rne(v4);
        const float r1 = v4 - bf16_to_f32(c0);
        uint32_t c1 = bf16_rne(r1);
        uint32_t c2 = bf16_rne(r1 - bf16_to_f32(c1));
        if (!safe) {
#pragma unroll
            for (int i = 0; i < 4; ++i) vh[i] = 0, vl[i] = 0;
            c0 = 0x7F80u, c1 = 0, c2 = 0;  // +inf * 1: -S* = +inf
        }
        // A operand of the threshold MFMA, element e of lane half h = slot 8h + e:
        //   vh0 vh0 vl0 vh1 vh1 vl1 c0 c1 | vh2 vh2 vl2 vh3 vh3 vl3 c2 0     against the candidate side's
        //   uh0 ul0 uh0 uh1 ul1 uh1  1  1 | uh2 ul2 uh2 uh3 ul3 uh3  1 0
        t[RQ_REC_V0 + 0] = vh[0] | (vh[0] << 16);
        t[RQ_REC_V0 + 1] = vl[0] | (vh[1] << 16);
        t[RQ_REC_V0 + 2] = vh[1] | (vl[1] << 16);
        t[RQ_REC_V0 + 3] = c0 | (c1 << 16);
        t[RQ_REC_V0 + 4] = vh[2] | (vh[2] << 16);
        t[RQ_REC_V0 + 5] = vl[2] | (vh[3] << 16);
        t[RQ_REC_V0 + 6] = vh[3] | (vl[3] << 16);
        t[RQ_REC_V0 + 7] = c2;
        // Dense run directory of a VALU stage: list position p of this pair lives in cell CELL0 + p / 64.  Cells follow
        // the stream: floor(stream_begin / 64) + 2 slot + 1 - floor(s_lo / 64) leaves every list its own cells (the two
        // spare cells per slot absorb the roundings of stream_begin and of the list's length) and never goes negative
        // for a position inside the stage.
        if (!tile_images) t[RQ_REC_CELL0] = (ps.stream_begin >> 6) + 2u * (p - ps.row * nprobe) + 1u - (s_lo >> 6);
        static_assert(RQ_REC_TAIL == 20, "five 16-byte pieces");
        uint4 piece = make_uint4(t[0], t[1], t[2], t[3]);  // 16-byte aligned in both layouts
#pragma unroll
        for (int i = 1; i < 5; ++i)
            if (sub == (uint32_t)i) piece = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        if (sub < 5) *reinterpret_cast<uint4 *>(tdst + 4 * sub) = piece;
    }
}
template <int LPP = 16>
__global__ __launch_bounds__(256) void stage_fill_kernel(const PairScalars *__restrict__ scal,
                                                         const uint32_t *__restrict__ probe_cluster,
                                                         const uint32_t *__restrict__ operand /* opdw dwords per pair */,
                                                         const float *__restrict__ thr, uint32_t count,
                                                         uint32_t nprobe, uint32_t slot_hi, uint32_t opdw, uint32_t s_lo,
                                                         uint32_t s_hi,
                                                         uint32_t cluster_major,
                                                         const uint32_t *__restrict__ grp_start,
                                                         uint32_t *__restrict__ grp_cursor,
                                                         uint32_t *__restrict__ recs, const FactorStats fs,
                                                         uint32_t tile_images,
                                                         const uint32_t *__restrict__ rank /* group_rank_kernel, or null */,
                                                         const uint32_t *__restrict__ blk_base, uint32_t k,
                                                         const float4 *__restrict__ list_uref,
                                                         const uint32_t *__restrict__ items = nullptr /* the work items to visit (count of them), else all */) {
    uint32_t wi = blockIdx.x * (256 / LPP) + threadIdx.x / LPP;  // work item: (query, slot < slot_hi)
    if (wi >= count) return;
    if (items) wi = items[wi];
    stage_fill_item<LPP>(scal, probe_cluster, operand, thr, wi, nprobe, slot_hi, opdw, s_lo, s_hi, cluster_major, grp_start, grp_cursor,
                    recs, fs, tile_images, rank, blk_base, k, list_uref);
}

// Additive gate of the matrix-core scan: centre V0 and half-range DV of v'_q = ((thr - ycd), -1, -lower, sqrt(ycd)) / (2 delta)
// over the pairs of a stage that probe list c (the records of group c, tile images of the additive format), written as
// two float4 per list.  Pairs whose start value is +inf (always flagged: stage_fill_kernel) take no part.  One block per list.
__global__ __launch_bounds__(256) void group_vrange_kernel(const uint32_t *__restrict__ recs, const uint32_t *__restrict__ grp_start,
                                                           const uint32_t *__restrict__ grp_cnt, uint32_t opdw,
                                                           float4 *__restrict__ vref) {
    const uint32_t c = blockIdx.x, n = grp_cnt[c], st0 = grp_start[c];
    const uint32_t opld = rq_img_opld(opdw, true), img = rq_img_dwords(opdw, true);
    const float inf = __builtin_inff();
    float lo[4] = {inf, inf, inf, inf}, hi[4] = {-inf, -inf, -inf, -inf};
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t at = st0 + i;
        const uint32_t *base = recs + (uint64_t)(at >> 5) * img + 32 * opld;
        const float cq = __builtin_bit_cast(float, base[32 * RQ_RECA_TAIL + (at & 31u)]);
        if (!(cq < inf)) continue;
        const uint32_t *t = base + (at & 31u) * RQ_RECA_TAIL;
        const float lower = __builtin_bit_cast(float, t[RQ_REC_LOWER]), delta = __builtin_bit_cast(float, t[RQ_REC_DELTA]);
        const float ycd = __builtin_bit_cast(float, t[RQ_REC_YCD]), ysq = __builtin_bit_cast(float, t[RQ_REC_YCD_SQRT]);
        const float th = __builtin_bit_cast(float, t[RQ_REC_THR]);
        const float inv2d = 0.5f / delta;  // the expressions of stage_fill_item
        const float v[4] = {(th - ycd) * inv2d, -inv2d, -lower * inv2d, ysq * inv2d};
#pragma unroll
        for (int r = 0; r < 4; ++r) lo[r] = fminf(lo[r], v[r]), hi[r] = fmaxf(hi[r], v[r]);
    }
    __shared__ float red[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        for (int o = 32; o >= 1; o >>= 1) lo[r] = fminf(lo[r], __shfl_xor(lo[r], o, 64)), hi[r] = fmaxf(hi[r], __shfl_xor(hi[r], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[threadIdx.x >> 6][r] = lo[r], red[threadIdx.x >> 6][4 + r] = hi[r];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float v0[4], dv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float l = fminf(fminf(red[0][r], red[1][r]), fminf(red[2][r], red[3][r]));
            const float u = fmaxf(fmaxf(red[0][4 + r], red[1][4 + r]), fmaxf(red[2][4 + r], red[3][4 + r]));
            const bool any = l <= u && u < inf && l > -inf;
            v0[r] = any ? 0.5f * l + 0.5f * u : 0.0f;
            // half-range, widened so that |v - v0| <= dv survives the roundings of v0 and of the subtraction
            dv[r] = any ? fmaxf(u - v0[r], v0[r] - l) * 1.000001f : 0.0f;
        }
        vref[2 * c] = make_float4(v0[0], v0[1], v0[2], v0[3]);
        vref[2 * c + 1] = make_float4(dv[0], dv[1], dv[2], dv[3]);
    }
}


// generic-W fallback (dim/64 not in the templated set): code words re-read per query (L1-resident)
__global__ __launch_bounds__(256) void scan_generic_kernel(SCAN_PARAMS, uint32_t W) {
    const uint32_t STRIDE = 8 * W + RQ_REC_TAIL;
    const uint32_t gl = blockIdx.x / a.tiles_per_group;
    const uint32_t g = a.group_base + gl;
    const uint32_t tile = a.tile_base + (blockIdx.x - gl * a.tiles_per_group);
    uint32_t pb, pe;
    if (a.cluster_major) {
        pb = grp_start[g];
        pe = grp_start[g + 1];
    } else {
        pb = g;
        pe = g + 1;
    }
    if (pb >= pe) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t *rec = recs + (uint64_t)pb * STRIDE;
    const uint32_t list_begin = rec[8 * W + RQ_REC_LIST_BEGIN], list_len = rec[8 * W + RQ_REC_LIST_LEN];
    const uint32_t first = tile * 256;
    if (first >= list_len) return;
    if (!a.cluster_major && rec[8 * W + RQ_REC_LO] >= rec[8 * W + RQ_REC_HI]) return;
    const uint32_t local = first + threadIdx.x;
    const uint32_t pos = list_begin + (local < list_len ? local : 0);
    const uint32_t *cp = codes + (uint64_t)pos * (2 * W);
    const float4 fac = factors[pos];
    for (uint32_t i = pb; i < pe; ++i, rec += STRIDE) {
        const uint32_t *pl = rec;  // the 4 bit planes (AND-popcount form)
        const uint32_t *t = rec + 8 * W;
        const uint32_t lo_p = t[RQ_REC_LO], hi_p = t[RQ_REC_HI];
        if (hi_p <= first || lo_p >= first + 256) continue;
        uint32_t s = 0;
        for (int pp = 0; pp < 4; ++pp) {
            uint32_t tt = 0;
            for (uint32_t w = 0; w < 2 * W; ++w) tt += __popc(cp[w] & pl[pp * 2 * W + w]);
            s += tt << pp;
        }
        float rough = rough_distance(s, fac, __builtin_bit_cast(float, t[RQ_REC_LOWER]),
                                     __builtin_bit_cast(float, t[RQ_REC_DELTA]), __builtin_bit_cast(float, t[RQ_REC_SUMQ]),
                                     __builtin_bit_cast(float, t[RQ_REC_YCD]), __builtin_bit_cast(float, t[RQ_REC_YCD_SQRT]));
        const uint32_t P0 = first + wave * 64;
        const uint32_t ra = lo_p > P0 ? (lo_p - P0 < 64 ? lo_p - P0 : 64) : 0;
        const uint32_t rb = hi_p > P0 ? (hi_p - P0 < 64 ? hi_p - P0 : 64) : 0;
        const uint64_t below_b = rb >= 64 ? ~0ull : ((1ull << rb) - 1ull);
        const uint64_t below_a = ra >= 64 ? ~0ull : ((1ull << ra) - 1ull);
        uint64_t m = __ballot(rough < __builtin_bit_cast(float, t[RQ_REC_THR])) & below_b & ~below_a;
        if (m) {
            const uint32_t b = t[RQ_REC_ROW], slot = t[RQ_REC_SLOT];
            const bool pass = (m >> lane) & 1ull;
            const uint32_t cntc = (uint32_t)__popcll(m);
            unsigned long long old = 0;
            if (lane == 0) old = atomicAdd(surv_cnt + b, (1ull << 32) | cntc);
            uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)old);
            uint32_t rbase = __builtin_amdgcn_readfirstlane((uint32_t)(old >> 32));
            if (pass) {
                uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (at < scan_seg(a).capof(b)) {
                    SurvRec r;
                    r.pos = pos, r.slot = slot, r.rough = rough, r.accurate = 0.0f;
                    surv[scan_seg(a).at(b) + at] = r;
                }
            }
            if (lane == 0 && rbase < scan_seg(a).capof(b)) {
                RunRec rr;
                rr.pos = list_begin + first + (threadIdx.x & ~63u);
                rr.slot = slot, rr.base = base, rr.cnt = cntc;
                runs[scan_seg(a).at(b) + rbase] = rr;
            }
        }
    }
}

// Dense variant for the per-stage test entry rq_scan: rough distance of every member of one list
// for one (query, list) pair; same device functions as the fused kernel.
__global__ __launch_bounds__(256) void scan_dense_kernel(const uint32_t *__restrict__ codes,
                                                         const float4 *__restrict__ factors,
                                                         uint32_t list_begin, uint32_t list_len,
                                                         uint32_t W, const uint32_t *__restrict__ pl,
                                                         float lower, float delta, float sumq, float ycd,
                                                         float *__restrict__ out) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= list_len) return;
    const uint32_t *cp = codes + (uint64_t)(list_begin + i) * (2 * W);
    uint32_t s = 0;
    for (int p = 0; p < 4; ++p) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < 2 * W; ++w) t += __popc(cp[w] & pl[p * 2 * W + w]);
        s += t << p;
    }
    out[i] = rough_distance(s, factors[list_begin + i], lower, delta, sumq, ycd, sqrtf(ycd));
}

// ------------------------------------------------------------------------------------------------
// Rerank distances (src/rerank.rs:85-90 -> src/simd.rs:14-73): accurate = ||base[pos] - q||^2 in
// the ORIGINAL space, exact AVX2 order.  8 GPU lanes play the 8 AVX lanes of one candidate (lane l
// runs the fused chain over elements 8c + l), so a wave reranks 8 survivors at a time and each
// 32-byte sector of the 4*dim-byte row is consumed by exactly one load.  In the query pipeline this
// is phase (A) of stage_finish_kernel.
// ------------------------------------------------------------------------------------------------
// flat variant for the per-stage test entry rq_rerank: positions given directly
__global__ __launch_bounds__(256) void accurate_flat_kernel(const uint32_t *__restrict__ pos, uint32_t m,
                                                            const BaseView base,
                                                            const float *__restrict__ q, uint32_t dim,
                                                            float *__restrict__ out) {
    const uint32_t l = threadIdx.x & 7;
    uint32_t i = blockIdx.x * 32 + (threadIdx.x >> 3);
    if (i >= m) return;
    const RowRef x = base.row(pos[i], dim);
    float acc = 0.0f;
    for (uint32_t c = 0; c < dim; c += 8) {
        float d = rq_row_get(x, dim, c + l) - q[c + l];
        acc = fmaf(d, d, acc);
    }
    acc = reduce8_lanes(acc);
    if (l == 0) out[i] = acc;
}

// ------------------------------------------------------------------------------------------------
// Restore the reference's visiting order among a query's survivors: ascending (slot, position)
// (src/rabitq.rs:304 outer loop, :348 inner loop).  One block per query; LDS when it fits.
// ------------------------------------------------------------------------------------------------
#define RQ_SORT_LDS_RECS 2048
template <typename T, uint32_t LDS_RECS = RQ_SORT_LDS_RECS>
__device__ __forceinline__ void sort_segment(T *recs, uint32_t n, const T *src = nullptr /* the unsorted records, when not in place */) {
    __shared__ T lds[LDS_RECS];
    if (!src) src = recs;
    if (n < 2 && src == recs) return;
    auto key = [](const T &r) { return surv_key(r); };
    if (n <= LDS_RECS) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds[i] = src[i];
        __syncthreads();
        bitonic_sort_block(lds, n, key);
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) recs[i] = lds[i];
    } else {
        if (src != recs) {
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) recs[i] = src[i];
            __threadfence_block();
        }
        __syncthreads();
        bitonic_sort_block(recs, n, key);  // in global memory (L2), rare
    }
}
// The run directory ordered by (slot, position) without a comparison sort over the whole directory: runs are
// bucketed by probe slot into a second buffer (LDS histogram + scatter: O(n)), then every bucket -- the runs one list
// contributed -- is ordered by position by ONE wave through rank counting (a run's final place is the number of
// smaller positions in its bucket; the bucket's positions are staged through LDS in chunks and compared four at a
// time) and written back to the directory at its final index.  Block-cooperative; needs nslots <= MAX_SLOTS.
#define RQ_BUCKET_CHUNK 1024u
#define RQ_SORT_MID_LDS_WORDS 14336u  // dwords of each of the two dynamic-LDS arrays of order_runs_bitmap: 458 752 cells = 14.7M list positions per query
template <uint32_t MAX_SLOTS>
__device__ __forceinline__ void sort_runs_by_slot(RunRec *__restrict__ dir, RunRec *__restrict__ tmp, uint32_t n, uint32_t nslots) {
    __shared__ uint32_t start[MAX_SLOTS + 1], cursor[MAX_SLOTS];
    __shared__ uint32_t wsum[16];
    __shared__ __attribute__((aligned(16))) uint32_t keys[4][RQ_BUCKET_CHUNK];  // per wave (blocks of 256 threads)
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    for (uint32_t i = tid; i < nslots; i += nthr) cursor[i] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nthr) atomicAdd(&cursor[dir[i].slot], 1u);
    __syncthreads();
    {  // exclusive scan of the bucket sizes: each thread owns a contiguous stretch of buckets
        const uint32_t per = (nslots + nthr - 1) / nthr, b0 = tid * per < nslots ? tid * per : nslots;
        const uint32_t b1 = b0 + per < nslots ? b0 + per : nslots;
        uint32_t sum = 0;
        for (uint32_t i = b0; i < b1; ++i) sum += cursor[i];
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wave; ++w) run += wsum[w];
        for (uint32_t i = b0; i < b1; ++i) {
            const uint32_t c = cursor[i];
            start[i] = run;
            cursor[i] = run;
            run += c;
        }
        if (tid == 0) start[nslots] = n;
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nthr) {
        const RunRec r = dir[i];
        tmp[atomicAdd(&cursor[r.slot], 1u)] = r;
    }
    __threadfence_block();
    __syncthreads();  // every record is in tmp (bucketed); dir is free to receive the final order
    for (uint32_t sl = wave; sl < nslots; sl += nwaves) {  // one wave per bucket
        const uint32_t b0 = start[sl], m = start[sl + 1] - b0;
        if (m == 0) continue;
        const RunRec *seg = tmp + b0;
        if (m == 1) {
            if (lane == 0) dir[b0] = seg[0];
            continue;
        }
        const bool one_chunk = m <= RQ_BUCKET_CHUNK;  // the usual case: the bucket's positions are staged once
        auto stage_keys = [&](uint32_t c0, uint32_t cm) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the previous contents have been consumed by every lane
            for (uint32_t t = lane; t < ((cm + 3) & ~3u); t += 64) keys[wave][t] = t < cm ? seg[c0 + t].pos : 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        if (one_chunk) stage_keys(0, m);
        for (uint32_t e0 = 0; e0 < m; e0 += 64) {  // 64 runs of the bucket at a time, one per lane
            RunRec mine;
            mine.pos = 0xFFFFFFFFu;
            if (e0 + lane < m) mine = seg[e0 + lane];
            uint32_t rank = 0;
            for (uint32_t c0 = 0; c0 < m; c0 += RQ_BUCKET_CHUNK) {
                const uint32_t cm = m - c0 < RQ_BUCKET_CHUNK ? m - c0 : RQ_BUCKET_CHUNK;
                if (!one_chunk) stage_keys(c0, cm);
                for (uint32_t t = 0; t < cm; t += 4) {
                    const uint4 kq = *reinterpret_cast<const uint4 *>(&keys[wave][t]);  // same address in every lane: broadcast
                    rank += (kq.x < mine.pos ? 1u : 0u) + (kq.y < mine.pos ? 1u : 0u) + (kq.z < mine.pos ? 1u : 0u) +
                            (kq.w < mine.pos ? 1u : 0u);
                }
            }
            if (e0 + lane < m) dir[b0 + rank] = mine;  // positions are unique within a bucket: ranks are a permutation
        }
    }
}

// The same ordering in O(n), for directories whose position span fits an LDS bitmap (the usual case: the final stage of a
// batch on an index with very unequal lists leaves tens of thousands of runs per query, thousands per list, where the rank
// counting above is quadratic).  Within a probe slot (= one list) the runs of one stage sit on distinct 32-position
// cells of the list (a run = one query x one 32- or 64-position sub-tile, common.h), so a run's final index is the number
// of occupied cells before its own: cell = cellbase[slot] + (pos - minpos[slot]) / 32 over a bitmap of the query's cells
// (one bit per 32 list positions of every probed list that contributed), rank = popcount prefix.  Three passes over the
// descriptors (L2), no comparison.  `src` are the unsorted descriptors, `out` receives the order (out != src).  Returns
// false -- nothing written -- when the bitmap does not fit `cap_words` or two runs share a cell (not produced by the
// scans; the caller then falls back to the bucket ranking).
template <uint32_t MAX_SLOTS>
__device__ __forceinline__ bool order_runs_bitmap(const RunRec *src, RunRec *out, uint32_t n, uint32_t nslots, uint32_t *words /* [cap_words] */,
                                                  uint32_t *wpre /* [cap_words] */, uint32_t cap_words) {
    __shared__ uint32_t minpos[MAX_SLOTS], cellbase[MAX_SLOTS];  // cellbase holds the slot's largest position first
    __shared__ uint32_t bsum[17];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < nslots; i += nthr) minpos[i] = 0xFFFFFFFFu, cellbase[i] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nthr) {
        const RunRec r = src[i];
        atomicMin(&minpos[r.slot], r.pos);
        atomicMax(&cellbase[r.slot], r.pos);
    }
    __syncthreads();
    // exclusive scan of the slots' cell counts (each thread owns a contiguous stretch of slots)
    uint32_t total_cells;
    {
        const uint32_t per = (nslots + nthr - 1) / nthr, b0 = tid * per < nslots ? tid * per : nslots;
        const uint32_t b1 = b0 + per < nslots ? b0 + per : nslots;
        uint32_t sum = 0;
        for (uint32_t i = b0; i < b1; ++i) sum += minpos[i] == 0xFFFFFFFFu ? 0u : ((cellbase[i] - minpos[i]) >> 5) + 1u;
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        if (lane == 63) bsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum, all = 0;
        for (uint32_t w = 0; w < (nthr >> 6); ++w) {
            if (w < wave) run += bsum[w];
            all += bsum[w];
        }
        total_cells = all;
        for (uint32_t i = b0; i < b1; ++i) {
            const uint32_t c = minpos[i] == 0xFFFFFFFFu ? 0u : ((cellbase[i] - minpos[i]) >> 5) + 1u;
            cellbase[i] = run;
            run += c;
        }
    }
    const uint32_t nwords = (total_cells + 31) >> 5;
    if (nwords > cap_words) return false;  // (block-uniform)
    for (uint32_t i = tid; i < nwords; i += nthr) words[i] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nthr) {
        const RunRec r = src[i];
        const uint32_t cell = cellbase[r.slot] + ((r.pos - minpos[r.slot]) >> 5);
        atomicOr(&words[cell >> 5], 1u << (cell & 31u));
    }
    __syncthreads();
    {  // exclusive popcount prefix over the words
        const uint32_t per = (nwords + nthr - 1) / nthr, w0 = tid * per < nwords ? tid * per : nwords;
        const uint32_t w1 = w0 + per < nwords ? w0 + per : nwords;
        uint32_t sum = 0;
        for (uint32_t i = w0; i < w1; ++i) sum += (uint32_t)__popc(words[i]);
        uint32_t incl = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        __syncthreads();  // bsum is free again
        if (lane == 63) bsum[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum, all = 0;
        for (uint32_t w = 0; w < (nthr >> 6); ++w) {
            if (w < wave) run += bsum[w];
            all += bsum[w];
        }
        if (all != n) return false;  // two runs on one cell (block-uniform)
        for (uint32_t i = w0; i < w1; ++i) {
            wpre[i] = run;
            run += (uint32_t)__popc(words[i]);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += nthr) {
        const RunRec r = src[i];
        const uint32_t cell = cellbase[r.slot] + ((r.pos - minpos[r.slot]) >> 5);
        out[wpre[cell >> 5] + (uint32_t)__popc(words[cell >> 5] & ((1u << (cell & 31u)) - 1u))] = r;
    }
    return true;
}

// heuristic ranker's accepted array (src/rerank.rs:170-176): by (Ord32(accurate), arrival)
__global__ __launch_bounds__(256) void sort_survivors_kernel(SurvRec *__restrict__ surv,
                                                             const uint32_t *__restrict__ surv_cnt,
                                                             uint32_t cap) {
    const uint32_t b = blockIdx.x;
    uint32_t n = surv_cnt[b];
    n = n < cap ? n : cap;
    sort_segment(surv + (uint64_t)b * cap, n);
}

// ------------------------------------------------------------------------------------------------
// Ordered replay of the re-rankers over (rough, accurate) pairs: HeapReRanker::rank_batch
// (src/rerank.rs:81-106) and HeuristicReRanker::rank_batch (:143-168), with Rust's
// std BinaryHeap push / pop (sift_up; sift_down_to_bottom + sift_up) on (Ord32, AlwaysEqual) items
// so that evictions among equal keys match.  One wave per query; survivors are taken 64 at a time
// and only lanes with rough < threshold are visited (ballot), the threshold being re-applied after
// every change.  State persists across stages in global memory.
// ------------------------------------------------------------------------------------------------
struct ReplayState {
    float *thr;              // nq
    uint32_t *heap_len;      // nq
    int32_t *heap_key;       // nq * topk
    uint32_t *heap_id;       // nq * topk
    uint32_t *precise;       // nq   (rerank.rs:91 / :153)
    uint32_t *need;          // nq   max survivor count of a stage (sizes re-runs and the learnt capacities)
    uint32_t *ovf;           // nq   1 = a stage dropped records (count > the query's capacity): the query is re-run
    uint32_t *nsurv;         // nq   survivors replayed (= accurate distances computed)
    uint32_t *nshadow;       // nq   of those: rejected by the fp16 shadow rows, f32 row never read
    // heuristic ranker
    float *recent_max;       // nq
    uint32_t *win_count;     // nq
    uint32_t *arr_len;       // nq   accepted so far (may exceed hcap -> overflow)
    SurvRec *arr;            // nq * hcap : {pos = arrival index, slot = biased Ord32(acc), accurate = id bits}
    uint32_t hcap;
};

#define RQ_MAX_TOPK 2048

// One wave replays a query's survivors (run directory `dir`, records `recs`) through the ranker.
// CONTIG: the survivors are recs[0 .. nruns) in visiting order already (no run directory: `dir` is unused and `nruns`
// counts records) -- the small-batch kernel's LDS-resident survivors.
template <bool HEURISTIC, bool REGHEAP = false, bool CONTIG = false>
__device__ __forceinline__ void replay_wave(const SurvRec *__restrict__ recs, const RunRec *__restrict__ dir,
                                            uint32_t nruns, uint32_t topk,
                                            uint32_t b, const ReplayState &st, int32_t *hkey, uint32_t *hid) {
    const uint32_t lane = threadIdx.x & 63;
    // Everything that steers the loops below is the same in every lane; saying so (v_readfirstlane / v_readlane) keeps the
    // loop counters, the heap indices and the branch conditions in scalar registers.  Left to the compiler's divergence
    // analysis, a bound that came out of a memory load or a cross-lane shuffle made the survivor loop "divergent" and with
    // it every value it carries: the sift loops then ran under exec masks with their indices in vector registers.
    nruns = __builtin_amdgcn_readfirstlane(nruns);
    topk = __builtin_amdgcn_readfirstlane(topk);
    float thr = st.thr[b];
    uint32_t precise = 0;
    uint32_t hlen = 0, wcount = 0, alen = 0;
    float recent = 0.0f;
    // REGHEAP (topk < 64: BinaryHeap::push before pop holds topk + 1 elements): the heap lives in one register pair, element i in lane i, read and written with
    // v_readlane / a lane-select at wave-uniform indices: a sift step is a few scalar instructions instead of a chain of
    // dependent LDS round trips (the replay of a stage was bound by exactly that latency)
    int32_t rk = 0;
    uint32_t ri = 0;
    auto HK = [&](uint32_t idx) -> int32_t {
        if constexpr (REGHEAP) return __builtin_amdgcn_readlane(rk, (int)idx);
        else return hkey[idx];
    };
    auto HI = [&](uint32_t idx) -> uint32_t {
        if constexpr (REGHEAP) return (uint32_t)__builtin_amdgcn_readlane((int)ri, (int)idx);
        else return hid[idx];
    };
    auto SETH = [&](uint32_t idx, int32_t k, uint32_t i) {
        if constexpr (REGHEAP) {
            rk = lane == idx ? k : rk;  // (no writelane builtin in this toolchain: a compare and two selects)
            ri = lane == idx ? i : ri;
        } else {
            hkey[idx] = k, hid[idx] = i;
        }
    };
    if constexpr (!HEURISTIC) {
        hlen = __builtin_amdgcn_readfirstlane(st.heap_len[b]);
        if constexpr (REGHEAP) {
            if (lane < hlen) rk = st.heap_key[(uint64_t)b * topk + lane], ri = st.heap_id[(uint64_t)b * topk + lane];
        } else {
            for (uint32_t i = lane; i < hlen; i += 64) {
                hkey[i] = st.heap_key[(uint64_t)b * topk + i];
                hid[i] = st.heap_id[(uint64_t)b * topk + i];
            }
        }
    } else {
        recent = st.recent_max[b];
        wcount = __builtin_amdgcn_readfirstlane(st.win_count[b]);
        alen = __builtin_amdgcn_readfirstlane(st.arr_len[b]);
    }
    // The survivors are replayed in stream order = directory order, then record order inside a run.  The
    // directory is read 64 descriptors at a time; within such a chunk the stream is cut into batches of 64
    // survivors (whatever runs they belong to): lane i finds its (run, offset) by a binary search over
    // the chunk's prefix sums in LDS, so a batch costs one round trip however many short runs it spans,
    // and batch k+1 is in flight while batch k is replayed.
    __shared__ uint32_t s_pref[CONTIG ? 1 : 65], s_base[CONTIG ? 1 : 64];
    for (uint32_t c0 = 0; c0 < (CONTIG ? (nruns ? 1u : 0u) : nruns); c0 += 64) {
        uint32_t total = nruns;
        if constexpr (!CONTIG) {
            uint32_t dbase = 0, dcnt = 0;
            if (c0 + lane < nruns) dbase = dir[c0 + lane].base, dcnt = dir[c0 + lane].cnt;
            uint32_t incl = dcnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o, 64);
                if ((int)lane >= o) incl += up;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // previous chunk's readers are done (one wave)
            s_pref[lane + 1] = incl;
            s_base[lane] = dbase;
            if (lane == 0) s_pref[0] = 0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        auto fetch = [&](uint32_t off, SurvRec &rec) {
            const uint32_t t = off + lane;
            rec.pos = 0, rec.slot = 0, rec.rough = 0.0f, rec.accurate = 0.0f;
            if (t < total) {
                if constexpr (CONTIG) {
                    rec = recs[t];
                } else {
                    uint32_t lo = 0;  // largest r with s_pref[r] <= t
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1)
                        if (lo + step < 64 && s_pref[lo + step] <= t) lo += step;
                    rec = recs[s_base[lo] + (t - s_pref[lo])];
                }
            }
        };
        SurvRec nxt;
        fetch(0, nxt);
        for (uint32_t off = 0; off < total; off += 64) {
            const bool have = off + lane < total;
            const SurvRec r = nxt;
            if (off + 64 < total) fetch(off + 64, nxt);  // wave-uniform
        uint64_t m = __ballot(have && r.rough < thr);  // rerank.rs:84 / :146: candidates the reference reranks
        while (m) {
            // the threshold only moves when a candidate is accepted (rerank.rs:92 / :154), so everything before
            // the next acceptance is counted in one step instead of visited one by one
            const uint64_t acc_m = __ballot(have && r.rough < thr && r.accurate < thr) & m;
            if (acc_m == 0) {
                precise += (uint32_t)__popcll(m);
                break;
            }
            const int i = __builtin_ctzll(acc_m);
            const uint64_t upto = (2ull << i) - 1ull;  // lanes 0..i (i = 63 wraps to all ones)
            precise += (uint32_t)__popcll(m & upto);
            m &= ~upto;
            // (lane i's values through v_readlane: i is wave-uniform, a shuffle would go through LDS)
            const float acc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.accurate), i));
            // the rankers carry the cluster-order POSITION; finalize_* maps it to the original id (rabitq.rs:324)
            const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)r.pos, i);
            if constexpr (!HEURISTIC) {
                // push: append + sift_up(0, old_len)
                int32_t key = ord32_from_f32(acc);
                uint32_t idv = id;
                if constexpr (REGHEAP) {  // every lane holds the same values: keep them (and the control flow) scalar
                    key = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)key);
                    idv = __builtin_amdgcn_readfirstlane(idv);
                }
                uint32_t p = hlen++;
                while (p > 0) {
                    uint32_t parent = (p - 1) >> 1;
                    int32_t pk = HK(parent);
                    if (key <= pk) break;
                    uint32_t pid = HI(parent);
                    SETH(p, pk, pid);
                    p = parent;
                }
                SETH(p, key, idv);
                if (hlen > topk) {  // pop: last -> root, sift_down_to_bottom(0), sift_up
                    --hlen;
                    int32_t hk = HK(hlen);
                    uint32_t hi = HI(hlen);
                    if (hlen > 0) {
                        const uint32_t end = hlen;
                        uint32_t q = 0, child = 1;
                        while (child + 1 < end) {
                            int32_t kl = HK(child), kr = HK(child + 1);
                            if (kl <= kr) child += 1;
                            int32_t ck = HK(child);
                            uint32_t ci = HI(child);
                            SETH(q, ck, ci);
                            q = child;
                            child = 2 * q + 1;
                        }
                        if (child == end - 1) {
                            int32_t ck = HK(child);
                            uint32_t ci = HI(child);
                            SETH(q, ck, ci);
                            q = child;
                        }
                        while (q > 0) {  // sift_up(0, q) of the hole element
                            uint32_t parent = (q - 1) >> 1;
                            int32_t pk = HK(parent);
                            if (hk <= pk) break;
                            uint32_t pid = HI(parent);
                            SETH(q, pk, pid);
                            q = parent;
                        }
                        SETH(q, hk, hi);
                    }
                }
                if (hlen == topk) thr = ord32_to_f32(HK(0));  // rerank.rs:98-100
            } else {
                if (alen < st.hcap && lane == 0) {
                    SurvRec e;
                    e.pos = alen;
                    e.slot = ord32_biased(acc);
                    e.rough = acc;
                    e.accurate = __builtin_bit_cast(float, id);
                    st.arr[(uint64_t)b * st.hcap + alen] = e;
                }
                ++alen;
                ++wcount;
                recent = (acc > recent || recent != recent) ? acc : recent;  // f32::max
                if (wcount >= 12) {                                          // consts.rs:12
                    thr = recent;
                    wcount = 0;
                    recent = -3.402823466e+38f;
                }
            }
            m &= __ballot(have && r.rough < thr);
        }
        }
    }
    if constexpr (!HEURISTIC) {
        if constexpr (REGHEAP) {
            if (lane < hlen) st.heap_key[(uint64_t)b * topk + lane] = rk, st.heap_id[(uint64_t)b * topk + lane] = ri;
        } else {
            for (uint32_t i = lane; i < hlen; i += 64) {
                st.heap_key[(uint64_t)b * topk + i] = hkey[i];
                st.heap_id[(uint64_t)b * topk + i] = hid[i];
            }
        }
        if (lane == 0) st.heap_len[b] = hlen;
    } else if (lane == 0) {
        st.recent_max[b] = recent;
        st.win_count[b] = wcount;
        st.arr_len[b] = alen;
    }
    if (lane == 0) {
        st.thr[b] = thr;
        st.precise[b] += precise;
    }
}

// exact f32 L2 of one SPLIT row (common.h: two 16-bit planes) against the query in LDS by a pair of lanes: the words are restored
// exactly, the arithmetic is accurate_rows' (lane half hf = AVX lanes 4hf..4hf+3, chunks of 64 dimensions in order)
__device__ __forceinline__ float exact_l2_pair_split(const float *__restrict__ row, const float *q_lds, uint32_t dim, uint32_t hf) {
    const uint16_t *hp = reinterpret_cast<const uint16_t *>(row) + 4 * hf, *lp = hp + dim;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (uint32_t c = 0; c < dim; c += 64) {
        uint2 hv[8], lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hv[u] = *reinterpret_cast<const uint2 *>(hp + c + 8 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) lv[u] = *reinterpret_cast<const uint2 *>(lp + c + 8 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 qv = *reinterpret_cast<const float4 *>(q_lds + c + 8 * u + 4 * hf);
            const float x0 = __builtin_bit_cast(float, (hv[u].x << 16) + (uint32_t)(int32_t)(int16_t)lv[u].x);
            const float x1 = __builtin_bit_cast(float, (hv[u].x & 0xFFFF0000u) + (uint32_t)((int32_t)lv[u].x >> 16));
            const float x2 = __builtin_bit_cast(float, (hv[u].y << 16) + (uint32_t)(int32_t)(int16_t)lv[u].y);
            const float x3 = __builtin_bit_cast(float, (hv[u].y & 0xFFFF0000u) + (uint32_t)((int32_t)lv[u].y >> 16));
            const float d0 = x0 - qv.x, d1 = x1 - qv.y, d2 = x2 - qv.z, d3 = x3 - qv.w;
            a0 = fmaf(d0, d0, a0), a1 = fmaf(d1, d1, a1), a2 = fmaf(d2, d2, a2), a3 = fmaf(d3, d3, a3);
        }
    }
    const float c0 = a0 + __shfl_xor(a0, 1, 2), c1 = a1 + __shfl_xor(a1, 1, 2);
    const float c2 = a2 + __shfl_xor(a2, 1, 2), c3 = a3 + __shfl_xor(a3, 1, 2);
    return (c0 + c1) + (c2 + c3);
}

// Exact f32 L2 of survivors recs[first], recs[first + step], ... against the query held in LDS
// (src/rerank.rs:85-90; lane order of src/simd.rs:14-73).  TWO lanes per candidate: lane half hf carries
// AVX lanes 4hf..4hf+3 (elements 8c + 4hf + 0..3, one 16-byte load per chunk); the fold
// ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)) needs one exchange between the two lanes.  A row is fetched 8
// chunks (64 dimensions, 8 x 16 bytes per lane) at a time so that every lane keeps 8 loads in flight
// (this is a random 512-byte-row gather: latency-bound unless enough bytes are outstanding).
__device__ __forceinline__ void accurate_rows(SurvRec *__restrict__ recs, uint32_t n, const BaseView &base,
                                              const float *q_lds, uint32_t dim, uint32_t first, uint32_t step,
                                              const uint32_t *__restrict__ probe_row /* the query's probed lists, by slot */) {
    const uint32_t hf = threadIdx.x & 1;
    for (uint32_t i = first; i < n; i += step) {
        const float *x;
        if (base.host == nullptr && !base.split) {
            x = base.dev + (uint64_t)recs[i].pos * dim + 4 * hf;
        } else {  // tiered: the survivor's slot names its list, the list's tier record places the row (HBM or host link)
            const RowRef rr = base.row_of_slot(recs[i].pos, probe_row, recs[i].slot, dim);
            if (rr.split) {  // (a pair of lanes shares its survivor: the branch does not split the pair)
                const float r = exact_l2_pair_split(rr.p, q_lds, dim, hf);
                if (hf == 0) recs[i].accurate = r;
                continue;
            }
            x = rr.p + 4 * hf;
        }
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (uint32_t c = 0; c < dim; c += 64) {  // dim is a multiple of 64
            float4 xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4 *>(x + c + 8 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 qv = *reinterpret_cast<const float4 *>(q_lds + c + 8 * u + 4 * hf);
                const float d0 = xv[u].x - qv.x, d1 = xv[u].y - qv.y, d2 = xv[u].z - qv.z, d3 = xv[u].w - qv.w;
                a0 = fmaf(d0, d0, a0), a1 = fmaf(d1, d1, a1), a2 = fmaf(d2, d2, a2), a3 = fmaf(d3, d3, a3);
            }
        }
        // c_i = a_i + a_{i+4}: the partner lane holds the other half (commutative, so both lanes agree)
        const float c0 = a0 + __shfl_xor(a0, 1, 2), c1 = a1 + __shfl_xor(a1, 1, 2);
        const float c2 = a2 + __shfl_xor(a2, 1, 2), c3 = a3 + __shfl_xor(a3, 1, 2);
        const float r = (c0 + c1) + (c2 + c3);
        if (hf == 0) recs[i].accurate = r;
    }
}

// One block per query finishes a stage: (A) exact rerank distances of the stage's survivors
// (src/rerank.rs:85-90, 8 lanes = the 8 AVX lanes of src/simd.rs:14-73), (B) sort of the run
// directory into the reference's visiting order, (C) wave 0 replays the ranker.
template <bool HEURISTIC>
__global__ __launch_bounds__(1024) void stage_finish_kernel(SurvRec *__restrict__ surv, RunRec *__restrict__ runs,
                                                           unsigned long long *__restrict__ surv_cnt, const QSeg seg,
                                                           const BaseView base,
                                                           const float *__restrict__ qpad, uint32_t dim, uint32_t topk,
                                                           ReplayState st, const uint32_t *__restrict__ probe_cluster,
                                                           uint32_t nprobe, uint32_t presorted) {
    __shared__ int32_t hkey[HEURISTIC ? 1 : RQ_MAX_TOPK];
    __shared__ uint32_t hid[HEURISTIC ? 1 : RQ_MAX_TOPK];
    extern __shared__ __attribute__((aligned(16))) float fin_q[];  // dim floats: the padded query
    const uint32_t b = blockIdx.x;
    const unsigned long long cnt64 = surv_cnt[b];
    const uint32_t cnt = (uint32_t)cnt64;
    const uint32_t cap = seg.capof(b);
    const uint64_t qat = seg.at(b);
    const bool overflow = cnt > cap;  // records were dropped: the query is re-run with a larger buffer
    const uint32_t n = overflow ? 0 : cnt;
    const uint32_t nruns = overflow ? 0 : (uint32_t)(cnt64 >> 32);
    __syncthreads();  // every thread has read the counter before thread 0 resets it
    if (threadIdx.x == 0) {
        if (cnt > st.need[b]) st.need[b] = cnt;
        if (overflow) st.ovf[b] = 1u;
        st.nsurv[b] += n;
        surv_cnt[b] = 0;  // ready for the next stage
    }
    if (n == 0) return;
    SurvRec *recs = surv + qat;
    {  // (A)
        for (uint32_t c = threadIdx.x * 4; c < dim; c += blockDim.x * 4)
            *reinterpret_cast<float4 *>(fin_q + c) = *reinterpret_cast<const float4 *>(qpad + (uint64_t)b * dim + c);
        __syncthreads();
        accurate_rows(recs, n, base, fin_q, dim, threadIdx.x >> 1, blockDim.x >> 1, probe_cluster + (uint64_t)b * nprobe);  // 256 or 1024 threads per query
    }
    // (B): up to RQ_SORT_LDS_RECS descriptors in LDS; longer directories were already ordered by sort_runs_mid_kernel
    // when the host launched it ahead of this kernel (presorted != 0), else (rare) bitonic in global memory
    if (nruns <= RQ_SORT_LDS_RECS || !presorted) sort_segment(runs + qat, nruns);
    __syncthreads();                                  // (A)'s stores and (B)'s order visible to wave 0
    if (threadIdx.x < 64)                             // (C)
        replay_wave<HEURISTIC>(recs, runs + qat, nruns, topk, b, st, hkey, hid);
}

// Rerank order of a large batch: queries grouped by their nearest list (counting sort: histogram, scan by
// group_scan_kernel, scatter).  Queries of one cluster rerank largely the same rows; handled back to back,
// the repeats are served by the L2 / Infinity Cache instead of HBM (measured: -15 % rerank time).
__global__ void order_count_kernel(const uint32_t *__restrict__ probe_cluster, uint32_t nprobe, uint32_t nq, uint32_t k,
                                   uint32_t *__restrict__ hist) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nq) return;
    const uint32_t c = probe_cluster[(uint64_t)b * nprobe];
    atomicAdd(&hist[c < k ? c : k], 1u);
}
__global__ void order_scatter_kernel(const uint32_t *__restrict__ probe_cluster, uint32_t nprobe, uint32_t nq, uint32_t k,
                                     const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor,
                                     uint32_t *__restrict__ order) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nq) return;
    const uint32_t c = probe_cluster[(uint64_t)b * nprobe];
    const uint32_t g = c < k ? c : k;
    order[start[g] + atomicAdd(&cursor[g], 1u)] = b;
}

// ------------------------------------------------------------------------------------------------
// Rerank pre-filter: an fp16 shadow of the raw vectors (derived state, half the bytes of a row).
//
// The re-ranker's exact distance only matters when it can pass `accurate < threshold` (src/rerank.rs:91): a
// survivor whose exact distance PROVABLY is >= the threshold its stage started with (the threshold only falls)
// is rejected by the reference whatever the exact value is.  For such a survivor the 4*dim-byte row is never
// fetched: the 2*dim-byte shadow row x~ gives
//     ||x - q||  >=  ||x~ - q|| - ||x - x~||,      ||x - x~|| <= 2^-11 ||x|| + sqrt(dim) 2^-25
// (fp16 round-to-nearest: relative 2^-11 per normal element, absolute 2^-25 per subnormal one; an element beyond
// the fp16 range becomes inf and disables the test), and the reference's f32 evaluation of ||x - q||^2
// (src/simd.rs:14-73: dim/8 fused multiply-adds per AVX lane + 3 adds, non-negative terms) is at least
// (1 - (dim/8 + 5) 2^-24) of the real value.  Every quantity below is rounded against the test (factors
// 1 -/+ eps with eps = (dim/4 + 64) 2^-24), so the bound can only be lower than the exact f32 result: a rejected
// survivor gets accurate = +inf, which fails `accurate < threshold` exactly as its exact value would.  Everything
// else goes through the exact path unchanged; results are bit-identical with and without the shadow.
// ------------------------------------------------------------------------------------------------
typedef _Float16 rq_half8 __attribute__((ext_vector_type(8)));

// 8 consecutive elements per thread; total = n * dim (a multiple of 64)
__global__ __launch_bounds__(256) void half_rows_kernel(const float *__restrict__ base, uint64_t total,
                                                        _Float16 *__restrict__ out) {
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < total; i += (uint64_t)gridDim.x * 2048) {
        const float4 a = *reinterpret_cast<const float4 *>(base + i), b = *reinterpret_cast<const float4 *>(base + i + 4);
        rq_half8 h;
        h[0] = (_Float16)a.x, h[1] = (_Float16)a.y, h[2] = (_Float16)a.z, h[3] = (_Float16)a.w;
        h[4] = (_Float16)b.x, h[5] = (_Float16)b.y, h[6] = (_Float16)b.z, h[7] = (_Float16)b.w;
        *reinterpret_cast<rq_half8 *>(out + i) = h;
    }
}

// exact f32 L2 of one row against the query in LDS by a PAIR of lanes (see accurate_rows); x already offset by 4*hf
__device__ __forceinline__ float exact_l2_pair(const float *__restrict__ x, const float *q_lds, uint32_t dim, uint32_t hf) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (uint32_t c = 0; c < dim; c += 64) {
        float4 xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4 *>(x + c + 8 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 qv = *reinterpret_cast<const float4 *>(q_lds + c + 8 * u + 4 * hf);
            const float d0 = xv[u].x - qv.x, d1 = xv[u].y - qv.y, d2 = xv[u].z - qv.z, d3 = xv[u].w - qv.w;
            a0 = fmaf(d0, d0, a0), a1 = fmaf(d1, d1, a1), a2 = fmaf(d2, d2, a2), a3 = fmaf(d3, d3, a3);
        }
    }
    const float c0 = a0 + __shfl_xor(a0, 1, 2), c1 = a1 + __shfl_xor(a1, 1, 2);
    const float c2 = a2 + __shfl_xor(a2, 1, 2), c3 = a3 + __shfl_xor(a3, 1, 2);
    return (c0 + c1) + (c2 + c3);
}

// grid (gx, nq), block 256 = 128 survivors per round, two lanes each.  Round: shadow-row test of 128 survivors; the
// ones it cannot reject queue up in LDS and are re-ranked exactly 128 at a time, so both phases keep every lane busy.
__global__ __launch_bounds__(256) void accurate_filtered_kernel(SurvRec *__restrict__ surv,
                                                                const unsigned long long *__restrict__ surv_cnt,
                                                                const QSeg seg, const float *__restrict__ base,
                                                                const _Float16 *__restrict__ base_h,
                                                                const float *__restrict__ qpad, uint32_t dim,
                                                                const uint32_t *__restrict__ order,
                                                                const float *__restrict__ thr_start,
                                                                uint32_t *__restrict__ nshadow) {
    extern __shared__ __attribute__((aligned(16))) float acc_q[];  // dim floats (the padded query)
    __shared__ uint32_t queue[256];
    __shared__ uint32_t qn;
    const uint32_t b = order ? order[blockIdx.y] : blockIdx.y;
    const uint32_t n = (uint32_t)surv_cnt[b];
    if (n > seg.capof(b) || n == 0) return;  // overflowed: this query is re-run with a larger buffer
    for (uint32_t c = threadIdx.x * 4; c < dim; c += 1024)
        *reinterpret_cast<float4 *>(acc_q + c) = *reinterpret_cast<const float4 *>(qpad + (uint64_t)b * dim + c);
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    SurvRec *recs = surv + seg.at(b);
    const uint32_t hf = threadIdx.x & 1, pair = threadIdx.x >> 1;
    const float thr = thr_start[b];
    const bool test = thr > 1e-30f && thr < 3.0e38f;  // a finite, normal threshold (false for NaN / inf: everything is exact)
    const float eps = (float)(dim / 4 + 64) * 5.9604645e-8f, down = 1.0f - eps, up = 1.0f + eps;
    const float abs_err = sqrtf((float)dim) * 3.0e-8f;  // sqrt(dim) * 2^-25, rounded up
    uint32_t rejected = 0;
    for (uint32_t i0 = blockIdx.x * 128; i0 < n; i0 += gridDim.x * 128) {
        const uint32_t i = i0 + pair;
        bool exact = i < n;
        if (exact && test) {
            const _Float16 *x = base_h + (uint64_t)recs[i].pos * dim + 8 * hf;
            float d0 = 0.0f, d1 = 0.0f, n0 = 0.0f, n1 = 0.0f;
            for (uint32_t c = 0; c < dim; c += 128) {  // 256 bytes of the row: 8 x 16 bytes per lane in flight
                rq_half8 xv[8];
                const bool full = dim - c >= 128;  // dim is a multiple of 64: the last chunk may be a half one
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < 4 || full) xv[u] = *reinterpret_cast<const rq_half8 *>(x + c + 16 * u);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (!(u < 4 || full)) continue;
                    const float4 qa = *reinterpret_cast<const float4 *>(acc_q + c + 16 * u + 8 * hf);
                    const float4 qb = *reinterpret_cast<const float4 *>(acc_q + c + 16 * u + 8 * hf + 4);
                    const float x0 = (float)xv[u][0], x1 = (float)xv[u][1], x2 = (float)xv[u][2], x3 = (float)xv[u][3];
                    const float x4 = (float)xv[u][4], x5 = (float)xv[u][5], x6 = (float)xv[u][6], x7 = (float)xv[u][7];
                    const float e0 = x0 - qa.x, e1 = x1 - qa.y, e2 = x2 - qa.z, e3 = x3 - qa.w;
                    const float e4 = x4 - qb.x, e5 = x5 - qb.y, e6 = x6 - qb.z, e7 = x7 - qb.w;
                    d0 = fmaf(e0, e0, d0), d1 = fmaf(e1, e1, d1), d0 = fmaf(e2, e2, d0), d1 = fmaf(e3, e3, d1);
                    d0 = fmaf(e4, e4, d0), d1 = fmaf(e5, e5, d1), d0 = fmaf(e6, e6, d0), d1 = fmaf(e7, e7, d1);
                    n0 = fmaf(x0, x0, n0), n1 = fmaf(x1, x1, n1), n0 = fmaf(x2, x2, n0), n1 = fmaf(x3, x3, n1);
                    n0 = fmaf(x4, x4, n0), n1 = fmaf(x5, x5, n1), n0 = fmaf(x6, x6, n0), n1 = fmaf(x7, x7, n1);
                }
            }
            float dt = d0 + d1, nx = n0 + n1;
            dt += __shfl_xor(dt, 1, 2), nx += __shfl_xor(nx, 1, 2);
#ifdef RQ_EXP_SHADOW_SLACK  // developer experiment (scripts/exp/shadow_slack.sh, rerank_shadow = 1): what a coarser shadow would still reject
            const float err = (sqrtf(nx) * up) * 4.8877e-4f + abs_err + RQ_EXP_SHADOW_SLACK;
#else
            const float err = (sqrtf(nx) * up) * 4.8877e-4f + abs_err;  // >= 2^-11 (1 + 2^-10) ||x~|| + sqrt(dim) 2^-25 >= ||x - x~||
#endif
            const float t = sqrtf(dt * down) * down - err * up;
            if (t > 0.0f && (t * t) * (down * down) > thr) {  // false for NaN
                exact = false;
                if (hf == 0) recs[i].accurate = __builtin_inff();
                ++rejected;
            }
        }
        if (exact && hf == 0) queue[atomicAdd(&qn, 1u)] = i;  // at most 127 waiting + 128 new
        __syncthreads();
        const uint32_t waiting = qn;  // the same value in every thread: nobody touches qn before the next barrier
        __syncthreads();
        if (waiting >= 128) {
            const uint32_t j = queue[waiting - 128 + pair];
            const float r = exact_l2_pair(base + (uint64_t)recs[j].pos * dim + 4 * hf, acc_q, dim, hf);
            if (hf == 0) recs[j].accurate = r;
            if (threadIdx.x == 0) qn = waiting - 128;
            __syncthreads();  // the queue's top 128 entries are free again, qn is set
        }
    }
    const uint32_t waiting = qn;
    if (pair < waiting) {
        const uint32_t j = queue[pair];
        const float r = exact_l2_pair(base + (uint64_t)recs[j].pos * dim + 4 * hf, acc_q, dim, hf);
        if (hf == 0) recs[j].accurate = r;
    }
    uint32_t rj = hf == 0 ? rejected : 0u;  // per-query counter (one address per query: no hot spot)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) rj += __shfl_xor(rj, o, 64);
    if ((threadIdx.x & 63) == 0 && rj) atomicAdd(&nshadow[b], rj);
}

#define RQ_ACC8_LDS_PROBES 1024u  // probe lists whose maps / tier records the re-rankers stage in LDS (16 B each)
// ------------------------------------------------------------------------------------------------
// Rerank over SPLIT rows (common.h: indexes whose raw vectors leave no room for shadow rows -- tiered ones, and untiered ones that
// fill most of the HBM): the
// first plane of a split row IS one: x^ = the row rounded to bf16, ||x - x^|| <= 2^-8 / (1 - 2^-8) ||x^|| (+ 2^-134 per subnormal
// element).  Same test as accurate_filtered_kernel's with that error term; a survivor it cannot reject has its second plane fetched
// and the words restored exactly, so results are bit-identical to the plain layout's.  Host-tier rows are split rows too (half
// the bytes over the host link for the ones the test rejects).  2*dim bytes per survivor instead of 4*dim for the ones the test rejects.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void accurate_split_kernel(SurvRec *__restrict__ surv,
                                                             const unsigned long long *__restrict__ surv_cnt,
                                                             const QSeg seg, const BaseView base,
                                                             const float *__restrict__ qpad, uint32_t dim,
                                                             const uint32_t *__restrict__ order,
                                                             const float *__restrict__ thr_start,
                                                             const uint32_t *__restrict__ probe_cluster, uint32_t nprobe,
                                                             uint32_t *__restrict__ nshadow) {
    extern __shared__ __attribute__((aligned(16))) float acc_q[];  // dim floats (the padded query), then -- when they fit -- the nprobe lists' tier records
    __shared__ uint32_t queue[256];
    __shared__ uint32_t qn;
    const uint32_t b = order ? order[blockIdx.y] : blockIdx.y;
    const uint32_t n = (uint32_t)surv_cnt[b];
    if (n > seg.capof(b) || n == 0) return;  // overflowed: this query is re-run with a larger buffer
    SurvRec *recs = surv + seg.at(b);
    const uint32_t *probe_row = probe_cluster + (uint64_t)b * nprobe;
    const uint32_t hf = threadIdx.x & 1, pair = threadIdx.x >> 1;
    // a chain of dependent gathers (survivor record -> its list's tier record -> the row): the first round's records are requested
    // before anything else, every later round's while the current one is worked on, the tier records are staged in LDS once per block
    const uint32_t i_first = blockIdx.x * 128;
    SurvRec nxt = recs[i_first + pair < n ? i_first + pair : 0u];
    ListTier *ltl = reinterpret_cast<ListTier *>(acc_q + dim);
    const bool lt_lds = base.host != nullptr && nprobe <= RQ_ACC8_LDS_PROBES;  // (launch: LDS for them only then; an untiered index has no tier records)
    for (uint32_t c = threadIdx.x * 4; c < dim; c += 1024)
        *reinterpret_cast<float4 *>(acc_q + c) = *reinterpret_cast<const float4 *>(qpad + (uint64_t)b * dim + c);
    if (lt_lds)
        for (uint32_t sl = threadIdx.x; sl < nprobe; sl += 256)  // (a padded probe slot -- id 0xFFFFFFFF, "no list" -- has no survivors and no record)
            ltl[sl] = probe_row[sl] < base.k ? base.lt[probe_row[sl]] : ListTier{0u, 0u, 0u, 0u};
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const float thr = thr_start ? thr_start[b] : __builtin_inff();
    const bool test = thr > 1e-30f && thr < 3.0e38f;  // a finite, normal threshold (false for NaN / inf: everything is exact)
    const float eps = (float)(dim / 4 + 64) * 5.9604645e-8f, down = 1.0f - eps, up = 1.0f + eps;
    auto row_of = [&](const SurvRec &r) { return lt_lds ? base.row_in_list(r.pos, ltl[r.slot], dim) : base.row_of_slot(r.pos, probe_row, r.slot, dim); };
    auto exact_of = [&](uint32_t j) {
        const RowRef rr = row_of(recs[j]);
        const float r = rr.split ? exact_l2_pair_split(rr.p, acc_q, dim, hf) : exact_l2_pair(rr.p + 4 * hf, acc_q, dim, hf);
        if (hf == 0) recs[j].accurate = r;
    };
    uint32_t rejected = 0;
    for (uint32_t i0 = i_first; i0 < n; i0 += gridDim.x * 128) {
        const uint32_t i = i0 + pair;
        bool exact = i < n;
        const SurvRec cur = nxt;
        {  // the next round's record (clamped: the last prefetch re-reads record 0)
            const uint32_t jn = i + gridDim.x * 128;
            nxt = recs[jn < n ? jn : 0u];
        }
        if (exact && test) {
            const RowRef rr = row_of(cur);
#ifdef RQ_EXP_COUNT_HOST  // developer experiment: the counter reports the host-tier survivors instead of the rejected ones
            if (!rr.split) ++rejected;
#endif
            if (rr.split) {
                const uint16_t *x = reinterpret_cast<const uint16_t *>(rr.p) + 8 * hf;
                float d0 = 0.0f, d1 = 0.0f, n0 = 0.0f, n1 = 0.0f;
                for (uint32_t c = 0; c < dim; c += 128) {  // 256 bytes of the plane: 8 x 16 bytes per lane in flight
                    uint4 xv[8];
                    const bool full = dim - c >= 128;  // dim is a multiple of 64: the last chunk may be a half one
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < 4 || full) xv[u] = *reinterpret_cast<const uint4 *>(x + c + 16 * u);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (!(u < 4 || full)) continue;
                        const float4 qa = *reinterpret_cast<const float4 *>(acc_q + c + 16 * u + 8 * hf);
                        const float4 qb = *reinterpret_cast<const float4 *>(acc_q + c + 16 * u + 8 * hf + 4);
                        const float x0 = __builtin_bit_cast(float, xv[u].x << 16), x1 = __builtin_bit_cast(float, xv[u].x & 0xFFFF0000u);
                        const float x2 = __builtin_bit_cast(float, xv[u].y << 16), x3 = __builtin_bit_cast(float, xv[u].y & 0xFFFF0000u);
                        const float x4 = __builtin_bit_cast(float, xv[u].z << 16), x5 = __builtin_bit_cast(float, xv[u].z & 0xFFFF0000u);
                        const float x6 = __builtin_bit_cast(float, xv[u].w << 16), x7 = __builtin_bit_cast(float, xv[u].w & 0xFFFF0000u);
                        const float e0 = x0 - qa.x, e1 = x1 - qa.y, e2 = x2 - qa.z, e3 = x3 - qa.w;
                        const float e4 = x4 - qb.x, e5 = x5 - qb.y, e6 = x6 - qb.z, e7 = x7 - qb.w;
                        d0 = fmaf(e0, e0, d0), d1 = fmaf(e1, e1, d1), d0 = fmaf(e2, e2, d0), d1 = fmaf(e3, e3, d1);
                        d0 = fmaf(e4, e4, d0), d1 = fmaf(e5, e5, d1), d0 = fmaf(e6, e6, d0), d1 = fmaf(e7, e7, d1);
                        n0 = fmaf(x0, x0, n0), n1 = fmaf(x1, x1, n1), n0 = fmaf(x2, x2, n0), n1 = fmaf(x3, x3, n1);
                        n0 = fmaf(x4, x4, n0), n1 = fmaf(x5, x5, n1), n0 = fmaf(x6, x6, n0), n1 = fmaf(x7, x7, n1);
                    }
                }
                float dt = d0 + d1, nx = n0 + n1;
                dt += __shfl_xor(dt, 1, 2), nx += __shfl_xor(nx, 1, 2);
                // >= 2^-8 / (1 - 2^-8) ||x^|| + sqrt(dim) 2^-134 >= ||x - x^||  (3.9216e-3 > 2^-8 / (1 - 2^-8) = 3.92157e-3; an
                // x^ with an infinite element makes err infinite and t NaN or -inf: no rejection)
                const float err = (sqrtf(nx) * up) * 3.9216e-3f + 1.0e-37f;
                const float t = sqrtf(dt * down) * down - err * up;
                if (t > 0.0f && (t * t) * (down * down) > thr) {  // false for NaN
                    exact = false;
                    if (hf == 0) recs[i].accurate = __builtin_inff();
#ifndef RQ_EXP_COUNT_HOST
                    ++rejected;
#endif
                }
            }
        }
        if (exact && hf == 0) queue[atomicAdd(&qn, 1u)] = i;  // at most 127 waiting + 128 new
        __syncthreads();
        const uint32_t waiting = qn;  // the same value in every thread: nobody touches qn before the next barrier
        __syncthreads();
        if (waiting >= 128) {
            exact_of(queue[waiting - 128 + pair]);
            if (threadIdx.x == 0) qn = waiting - 128;
            __syncthreads();  // the queue's top 128 entries are free again, qn is set
        }
    }
    const uint32_t waiting = qn;
    if (pair < waiting) exact_of(queue[pair]);
    uint32_t rj = hf == 0 ? rejected : 0u;  // per-query counter (one address per query: no hot spot)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) rj += __shfl_xor(rj, o, 64);
    if ((threadIdx.x & 63) == 0 && rj) atomicAdd(&nshadow[b], rj);
}

// ------------------------------------------------------------------------------------------------
// The 8-bit shadow (round 4): one BYTE per dimension instead of the fp16 shadow's two.
//
// Every list c has an affine map of its own, x^_i = lo_c + s_c * code_i with lo_c = the smallest and lo_c + 255 s_c = the largest
// coordinate of any of its rows (q8_range_kernel), so no coordinate clips and |x_i - x^_i| <= s_c / 2 up to rounding.  The bound
// the pre-filter needs, ||x - x^|| over the rows of the list, is not derived but MEASURED while the codes are written
// (q8_encode_kernel: the largest row norm of x - fmaf(s_c, code, lo_c) over the list, x^ evaluated exactly as the re-ranker evaluates
// it -- round 5; for dimensions whose rows do not map onto a power-of-two thread group: the largest |x_i - x^_i| times sqrt(dim)).
// The test itself is accurate_filtered_kernel's: d^ = ||x^ - q|| in f32, t = d^ (1 - eps) - err, and a survivor is
// dropped only if t^2 (1 - eps) still exceeds the stage's threshold -- the f32 row is then never read (128 instead of 256 shadow
// bytes per survivor at dim 128; measured on the benchmark mixture: 86 % of the survivors rejected against the fp16 shadow's 92 %).
// A list with a non-finite coordinate gets err = inf: nothing of it is ever rejected.
// ------------------------------------------------------------------------------------------------
// whether q8_encode_kernel measures row norms of the error (list_q8[c].w) for this dimension: its dim / 16 threads per row must be
// a power-of-two group inside one wave
__host__ __device__ __forceinline__ bool q8_row_norms(uint32_t dim) {
    const uint32_t tpr = dim / 16;
    return tpr >= 1 && tpr <= 64 && (tpr & (tpr - 1)) == 0;
}
// one block per list: lo, s (and the error accumulator cleared)
__global__ __launch_bounds__(256) void q8_range_kernel(const float *__restrict__ base, const uint32_t *__restrict__ offsets, uint32_t dim,
                                                       float4 *__restrict__ list_q8) {
    __shared__ float smn[4], smx[4];
    __shared__ uint32_t sbad[4];
    const uint32_t c = blockIdx.x;
    const uint64_t e0 = (uint64_t)offsets[c] * dim, e1 = (uint64_t)offsets[c + 1] * dim;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    uint32_t bad = 0;
    for (uint64_t e = e0 + threadIdx.x * 4ull; e < e1; e += 1024) {  // dim is a multiple of 64: rows are whole float4s
        const float4 v = *reinterpret_cast<const float4 *>(base + e);
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!(fabsf(ve[i]) < 3.0e38f)) bad = 1;  // NaN / inf / huge
            mn = ve[i] < mn ? ve[i] : mn;
            mx = ve[i] > mx ? ve[i] : mx;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64);
        mn = a < mn ? a : mn, mx = b > mx ? b : mx;
        bad |= __shfl_xor(bad, o, 64);
    }
    if ((threadIdx.x & 63) == 0) smn[threadIdx.x >> 6] = mn, smx[threadIdx.x >> 6] = mx, sbad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) mn = smn[w] < mn ? smn[w] : mn, mx = smx[w] > mx ? smx[w] : mx, bad |= sbad[w];
        float lo = mn, sc = (mx - mn) * (1.0f / 255.0f);
        if (e1 == e0) lo = 0.0f, sc = 1.0f;
        if (!(sc > 0.0f)) sc = 1.0f;                       // all coordinates equal (or an empty list): code 0 everywhere
        if (!(sc < 3.0e38f) || !(fabsf(lo) < 3.0e38f)) bad = 1;
        // .z: the largest |x_i - x^_i| of the list as the bits of a non-negative float (atomicMax by q8_encode_kernel); inf: never reject
        list_q8[c] = make_float4(bad ? 0.0f : lo, bad ? 1.0f : sc, bad ? __builtin_inff() : 0.0f, bad ? __builtin_inff() : 0.0f);  // .w: the largest row norm of the error, likewise
    }
}
// grid (ceil(longest list / 256), k): 256 rows of one list per block; dim / 16 threads per row, 16 codes (one 16-byte store) each
__global__ __launch_bounds__(256) void q8_encode_kernel(const float *__restrict__ base, const uint32_t *__restrict__ offsets, uint32_t dim,
                                                        float4 *__restrict__ list_q8, uint8_t *__restrict__ out) {
    __shared__ float smax[4], snorm[4];
    const uint32_t c = blockIdx.y, r0 = offsets[c] + blockIdx.x * 256u, r1 = offsets[c + 1];
    if (r0 >= r1) return;
    const float4 par = list_q8[c];
    const float lo = par.x, sc = par.y, inv = 1.0f / sc;
    const uint32_t tpr = dim / 16, rows_per_pass = 256 / tpr;  // dim <= 4096
    const uint32_t rr = threadIdx.x / tpr, g = threadIdx.x - rr * tpr;
    const bool row_norms = q8_row_norms(dim);  // the tpr threads of a row are an aligned power-of-two group of one wave
    float emax = 0.0f, nmax = 0.0f;
    for (uint32_t r = r0 + rr; r < r1 && r < r0 + 256u && rr < rows_per_pass; r += rows_per_pass) {
        float e2 = 0.0f;
        const float *x = base + (uint64_t)r * dim + 16 * g;
        uint32_t w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 v = *reinterpret_cast<const float4 *>(x + 4 * u);
            const float ve[4] = {v.x, v.y, v.z, v.w};
            w[u] = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = rintf((ve[i] - lo) * inv);
                t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);  // (NaN: a list with one has err = inf already)
                const uint32_t code = (uint32_t)t;
                const float e = fabsf(ve[i] - fmaf(sc, (float)code, lo));  // exactly the re-ranker's x^
                emax = (e > emax || !(e < 3.0e38f)) ? (e < 3.0e38f ? e : __builtin_inff()) : emax;
                e2 = fmaf(e, e, e2);
                w[u] |= code << (8 * i);
            }
        }
        *reinterpret_cast<uint4 *>(out + (uint64_t)r * dim + 16 * g) = make_uint4(w[0], w[1], w[2], w[3]);
        if (row_norms) {  // ||x - x^||^2 of the row: the group's partial sums folded (every thread of the group is in this iteration)
            for (uint32_t o = tpr >> 1; o >= 1; o >>= 1) e2 += __shfl_xor(e2, (int)o, 64);
            nmax = (e2 > nmax || !(e2 < 3.0e38f)) ? (e2 < 3.0e38f ? e2 : __builtin_inff()) : nmax;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float a = __shfl_xor(emax, o, 64), b = __shfl_xor(nmax, o, 64);
        emax = a > emax ? a : emax, nmax = b > nmax ? b : nmax;
    }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = emax, snorm[threadIdx.x >> 6] = nmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w2 = 1; w2 < 4; ++w2) emax = smax[w2] > emax ? smax[w2] : emax, nmax = snorm[w2] > nmax ? snorm[w2] : nmax;
        atomicMax(reinterpret_cast<unsigned int *>(&list_q8[c].z), __builtin_bit_cast(unsigned int, emax));  // non-negative floats order as their bits
        // .w: the largest ||x - x^|| over the list's rows, rounded up past the f32 evaluation above (dim fused multiply-adds + a sqrt)
        const float nrm = sqrtf(nmax) * (1.0f + (float)(dim + 8) * 6.0e-8f);
        if (row_norms) atomicMax(reinterpret_cast<unsigned int *>(&list_q8[c].w), __builtin_bit_cast(unsigned int, nrm));
    }
}

// accurate_filtered_kernel with the 8-bit shadow: grid (gx, nq), block 256 = 256 survivors per round, two lanes each and two
// survivors per lane pair (lane half hf takes the 16-dimension groups 2u + hf of a row; both survivors' pieces are requested
// before either is used: a 128-byte row is four 16-byte loads per lane, too few in flight to keep the memory busy one row at a time)
#ifndef RQ_ACC8_WAVES
#define RQ_ACC8_WAVES 4  // waves per SIMD the register budget is cut for (round 4: 132 registers, three waves; four: rerank 4.76 -> 4.35 ms per step)
#endif
__global__ __launch_bounds__(256, RQ_ACC8_WAVES) void accurate_filtered8_kernel(SurvRec *__restrict__ surv,
                                                                 const unsigned long long *__restrict__ surv_cnt,
                                                                 const QSeg seg, const float *__restrict__ base,
                                                                 const uint8_t *__restrict__ base_q8, const float4 *__restrict__ list_q8,
                                                                 const float *__restrict__ qpad, uint32_t dim,
                                                                 const uint32_t *__restrict__ order,
                                                                 const float *__restrict__ thr_start,
                                                                 const uint32_t *__restrict__ probe_cluster, uint32_t nprobe,
                                                                 uint32_t *__restrict__ nshadow, uint32_t nlists) {
    extern __shared__ __attribute__((aligned(16))) float acc_q[];  // dim floats (the padded query), then -- when they fit -- the nprobe lists' (lo, s, err)
    __shared__ uint32_t queue[512];
    __shared__ uint32_t qn;
    const uint32_t b = order ? order[blockIdx.y] : blockIdx.y;
    const uint32_t n = (uint32_t)surv_cnt[b];
    if (n > seg.capof(b) || n == 0) return;  // overflowed: this query is re-run with a larger buffer
    SurvRec *recs = surv + seg.at(b);
    const uint32_t hf = threadIdx.x & 1, pair = threadIdx.x >> 1;
    // The kernel is a chain of dependent gathers (survivor record -> its list's map and its shadow row -> the f32 row of the few that
    // stay), i.e. bound by how many of them are in flight: the first round's records are requested before anything else, every later
    // round's while the current one is being worked on, and the lists' maps are staged in LDS once per block (one dependent round
    // trip less per round: probe list -> map was two)
    const uint32_t i_first = blockIdx.x * 256;
    SurvRec nxt[2];
    nxt[0] = recs[i_first + pair < n ? i_first + pair : 0u], nxt[1] = recs[i_first + 128 + pair < n ? i_first + 128 + pair : 0u];
    const uint32_t *pc = probe_cluster + (uint64_t)b * nprobe;
    float4 *lpar = reinterpret_cast<float4 *>(acc_q + dim);
    const bool par_lds = nprobe <= RQ_ACC8_LDS_PROBES;  // (launch: LDS for them only then)
    for (uint32_t c = threadIdx.x * 4; c < dim; c += 1024)
        *reinterpret_cast<float4 *>(acc_q + c) = *reinterpret_cast<const float4 *>(qpad + (uint64_t)b * dim + c);
    if (par_lds)
        for (uint32_t sl = threadIdx.x; sl < nprobe; sl += 256)  // (a padded probe slot -- id 0xFFFFFFFF, "no list" -- has no survivors and no map)
            lpar[sl] = pc[sl] < nlists ? list_q8[pc[sl]] : make_float4(0.0f, 1.0f, __builtin_inff(), __builtin_inff());
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const float thr = thr_start[b];
    const bool test = thr > 1e-30f && thr < 3.0e38f;  // a finite, normal threshold (false for NaN / inf: everything is exact)
    const float eps = (float)(dim / 4 + 64) * 5.9604645e-8f, down = 1.0f - eps, up = 1.0f + eps;
    const float sqd = sqrtf((float)dim) * 1.001f;
    const bool rown = q8_row_norms(dim);
    const uint32_t ngrp = dim / 16;
    uint32_t rejected = 0;
    for (uint32_t i0 = i_first; i0 < n; i0 += gridDim.x * 256) {
        uint32_t iv[2] = {i0 + pair, i0 + 128 + pair};
        bool exact[2] = {iv[0] < n, iv[1] < n};
        const SurvRec cur[2] = {nxt[0], nxt[1]};
        {  // the next round's records (clamped: the last prefetch re-reads record 0)
            const uint32_t j0 = i0 + gridDim.x * 256 + pair, j1 = j0 + 128;
            nxt[0] = recs[j0 < n ? j0 : 0u], nxt[1] = recs[j1 < n ? j1 : 0u];
        }
        if (test) {
            float4 par[2];
            const uint8_t *x[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                par[v] = par_lds ? lpar[cur[v].slot] : list_q8[pc[cur[v].slot]];  // lo, s, max |x_i - x^_i| of the list
                x[v] = base_q8 + (uint64_t)cur[v].pos * dim;
            }
            float d0[2] = {0.0f, 0.0f}, d1[2] = {0.0f, 0.0f};
            for (uint32_t g0 = hf; g0 < ngrp; g0 += 8) {  // four 16-byte pieces of each row in flight per lane
                uint4 cv[2][4];
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (g0 + 2 * u < ngrp) cv[v][u] = *reinterpret_cast<const uint4 *>(x[v] + 16 * (g0 + 2 * u));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!(g0 + 2 * u < ngrp)) continue;
                    const float *q = acc_q + 16 * (g0 + 2 * u);
#pragma unroll
                    for (int wi = 0; wi < 4; ++wi) {
                        const float4 qa = *reinterpret_cast<const float4 *>(q + 4 * wi);
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const uint32_t wv = wi == 0 ? cv[v][u].x : (wi == 1 ? cv[v][u].y : (wi == 2 ? cv[v][u].z : cv[v][u].w));
                            const float e0 = fmaf(par[v].y, (float)(wv & 255u), par[v].x) - qa.x;
                            const float e1 = fmaf(par[v].y, (float)((wv >> 8) & 255u), par[v].x) - qa.y;
                            const float e2 = fmaf(par[v].y, (float)((wv >> 16) & 255u), par[v].x) - qa.z;
                            const float e3 = fmaf(par[v].y, (float)(wv >> 24), par[v].x) - qa.w;
                            d0[v] = fmaf(e0, e0, d0[v]), d1[v] = fmaf(e1, e1, d1[v]), d0[v] = fmaf(e2, e2, d0[v]), d1[v] = fmaf(e3, e3, d1[v]);
                        }
                    }
                    // (the query pieces of the later groups are read when their turn comes: hoisted, the sixteen 16-byte LDS reads of
                    // a round held 64 registers and cost the kernel a wave per SIMD)
                    asm volatile("" ::: "memory");
                }
            }
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                float dt = d0[v] + d1[v];
                dt += __shfl_xor(dt, 1, 2);
                // >= ||x - x^||: the largest row norm of the error measured over the list (dimensions q8_encode_kernel measures it for),
                // else sqrt(dim) max |x_i - x^_i|   (inf: a list that is never rejected)
                const float err = rown ? par[v].w * up : (par[v].z * up) * sqd;
                const float t = sqrtf(dt * down) * down - err * up;
                if (exact[v] && t > 0.0f && (t * t) * (down * down) > thr) {  // false for NaN
                    exact[v] = false;
                    if (hf == 0) recs[iv[v]].accurate = __builtin_inff();
                    ++rejected;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 2; ++v)
            if (exact[v] && hf == 0) queue[atomicAdd(&qn, 1u)] = iv[v];  // at most 127 waiting + 256 new
        __syncthreads();
        uint32_t waiting = qn;  // the same value in every thread: nobody touches qn before the next barrier
        __syncthreads();
        while (waiting >= 128) {  // block-uniform
            const uint32_t j = queue[waiting - 128 + pair];
            const float r = exact_l2_pair(base + (uint64_t)recs[j].pos * dim + 4 * hf, acc_q, dim, hf);
            if (hf == 0) recs[j].accurate = r;
            waiting -= 128;
            __syncthreads();  // the queue's top 128 entries are free again
            if (threadIdx.x == 0) qn = waiting;
            __syncthreads();
        }
    }
    const uint32_t waiting = qn;
    if (pair < waiting) {
        const uint32_t j = queue[pair];
        const float r = exact_l2_pair(base + (uint64_t)recs[j].pos * dim + 4 * hf, acc_q, dim, hf);
        if (hf == 0) recs[j].accurate = r;
    }
    uint32_t rj = hf == 0 ? rejected : 0u;  // per-query counter (one address per query: no hot spot)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) rj += __shfl_xor(rj, o, 64);
    if ((threadIdx.x & 63) == 0 && rj) atomicAdd(&nshadow[b], rj);
}

// ---- the same three phases as separate launches: better for large batches, where all queries'
// survivors are reranked with full-chip parallelism before the (latency-bound) replay --------------
// grid (gx, nq); block 256
__global__ __launch_bounds__(256) void accurate_kernel(SurvRec *__restrict__ surv,
                                                       const unsigned long long *__restrict__ surv_cnt,
                                                       const QSeg seg, const BaseView base,
                                                       const float *__restrict__ qpad, uint32_t dim,
                                                       const uint32_t *__restrict__ order,
                                                       const uint32_t *__restrict__ probe_cluster, uint32_t nprobe) {
    // TWO lanes per candidate: lane half hf carries AVX lanes 4hf..4hf+3 (elements 8c + 4hf + 0..3, one
    // 16-byte load per chunk), so a row is fetched with float4 loads; the fold
    // ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)) needs one exchange between the two lanes.
    extern __shared__ __attribute__((aligned(16))) float acc_q[];  // dynamic LDS: dim floats (the padded query)
    const uint32_t b = order ? order[blockIdx.y] : blockIdx.y;  // consecutive blocks: queries of the same nearest list
    const uint32_t n = (uint32_t)surv_cnt[b];
    if (n > seg.capof(b) || n == 0) return;  // overflowed: this query is re-run with a larger buffer
    for (uint32_t c = threadIdx.x * 4; c < dim; c += 1024)
        *reinterpret_cast<float4 *>(acc_q + c) = *reinterpret_cast<const float4 *>(qpad + (uint64_t)b * dim + c);
    __syncthreads();
    accurate_rows(surv + seg.at(b), n, base, acc_q, dim, blockIdx.x * 128 + (threadIdx.x >> 1), gridDim.x * 128,
                  probe_cluster + (uint64_t)b * nprobe);
}

// `runs_src`: where the stage's unsorted descriptors are when not in `runs` itself (an arena stage scatters them into the
// second directory buffer, so that the ordering pass is the one that writes the directory); same geometry.
__global__ __launch_bounds__(64) void sort_runs_kernel(RunRec *__restrict__ runs,
                                                        const unsigned long long *__restrict__ surv_cnt,
                                                        const QSeg seg, uint32_t *__restrict__ big_list,
                                                        uint32_t *__restrict__ big_count, uint32_t list_above,
                                                        const RunRec *runs_src) {
    const uint32_t b = blockIdx.x;
    const unsigned long long c = surv_cnt[b];
    if ((uint32_t)c > seg.capof(b)) return;
    // early stages leave a few dozen runs per query, the stages around one list's worth a few hundred (more at dim 64,
    // where the estimates are noisier): 512 descriptors = 8 KiB of LDS per 64-thread block keep them out of global memory
    const uint32_t nruns = (uint32_t)(c >> 32);
    if (list_above != 512u) {  // small-batch path: only list the directories stage_finish_kernel cannot sort in LDS
        if (nruns > list_above && threadIdx.x == 0) big_list[atomicAdd(big_count, 1u)] = b;
        return;
    }
    if (nruns > 512) {  // a loose threshold: handed to sort_runs_mid_kernel (cell bitmap, or slot buckets + rank counting)
        if (threadIdx.x == 0) big_list[atomicAdd(big_count, 1u)] = b;
        return;
    }
    sort_segment<RunRec, 512>(runs + seg.at(b), nruns, runs_src ? runs_src + seg.at(b) : nullptr);
}

// Directories of more than 512 runs, listed by sort_runs_kernel, are ordered by a persistent launch that walks the
// list (it exits at once when the list is empty, the common case): the cell bitmap (order_runs_bitmap, dynamic LDS:
// 2 x lds_words dwords), else slot-bucketing + per-bucket rank counting through the second directory buffer; the last
// block out resets the counter for the next stage.  src_is_tmp: the unsorted descriptors are in runs_tmp.
__global__ __launch_bounds__(256) void sort_runs_mid_kernel(RunRec *__restrict__ runs, RunRec *__restrict__ runs_tmp,
                                                            const unsigned long long *__restrict__ surv_cnt, const QSeg seg,
                                                            const uint32_t *__restrict__ big_list,
                                                            uint32_t *__restrict__ big_count /* [0] entries, [1] blocks done, [2] most entries of a stage */,
                                                            uint32_t nslots, uint32_t src_is_tmp, uint32_t lds_words) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mid_lds[];
    const uint32_t total = big_count[0];
    for (uint32_t i = blockIdx.x; i < total; i += gridDim.x) {
        const uint32_t b = big_list[i], n = (uint32_t)(surv_cnt[b] >> 32);
        RunRec *dir = runs + seg.at(b), *tmp = runs_tmp ? runs_tmp + seg.at(b) : nullptr;
        bool done = false;
        if (nslots <= 1024 && tmp && lds_words) {
            if (src_is_tmp) {
                done = order_runs_bitmap<1024>(tmp, dir, n, nslots, mid_lds, mid_lds + lds_words, lds_words);
            } else {
                done = order_runs_bitmap<1024>(dir, tmp, n, nslots, mid_lds, mid_lds + lds_words, lds_words);
                if (done) {  // back into the directory (the block's own writes: L2)
                    __threadfence_block();
                    __syncthreads();
                    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) dir[e] = tmp[e];
                }
            }
        }
        if (!done) {
            if (src_is_tmp && tmp) {  // the fall-backs order the directory itself
                __syncthreads();
                for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) dir[e] = tmp[e];
                __threadfence_block();
                __syncthreads();
            }
            if (nslots <= 1024 && tmp) sort_runs_by_slot<1024>(dir, tmp, n, nslots);
            else sort_segment<RunRec, 16>(dir, n);  // more than 1024 probe slots, or no second buffer yet: bitonic sort in global memory
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // every block reads `total` before it counts itself done: the last one out may reset both
        __threadfence();
        if (atomicAdd(big_count + 1, 1u) + 1 == gridDim.x) {
            if (total > big_count[2]) big_count[2] = total;  // for the host: sizes the next pass's launch
            big_count[0] = 0;
            big_count[1] = 0;
        }
    }
}

template <bool HEURISTIC, bool REGHEAP = false>
__global__ __launch_bounds__(64) void replay_kernel(const SurvRec *__restrict__ surv, const RunRec *__restrict__ runs,
                                                    unsigned long long *__restrict__ surv_cnt, const QSeg seg, uint32_t topk,
                                                    ReplayState st, uint32_t dense_cells) {
    extern __shared__ __attribute__((aligned(16))) unsigned char replay_smem[];  // topk * 8 bytes (heap ranker)
    int32_t *hkey = reinterpret_cast<int32_t *>(replay_smem);
    uint32_t *hid = reinterpret_cast<uint32_t *>(replay_smem) + topk;
    const uint32_t b = blockIdx.x;
    const unsigned long long cnt64 = surv_cnt[b];
    const uint32_t cnt = (uint32_t)cnt64;
    const bool overflow = cnt > seg.capof(b);
    const uint32_t n = overflow ? 0 : cnt;
    // dense directory: every cell of the stage is a descriptor (count 0 where nothing survived), already in order
    const uint32_t nruns = overflow ? 0 : (dense_cells ? dense_cells : (uint32_t)(cnt64 >> 32));
    if (threadIdx.x == 0) {
        if (cnt > st.need[b]) st.need[b] = cnt;
        if (overflow) st.ovf[b] = 1u;
        st.nsurv[b] += n;
        surv_cnt[b] = 0;  // ready for the next stage
    }
    if (n == 0) return;
    replay_wave<HEURISTIC, REGHEAP>(surv + seg.at(b), runs + seg.at(b), nruns, topk, b, st, hkey, hid);
}

// Segment sizing of an arena stage: query b's segment = its exact survivor count (surv_cnt low word, counted by the scan)
// rounded up to 64 slots, at least floor_cap.
__global__ void seg_exact_kernel(unsigned long long *__restrict__ surv_cnt, uint32_t nq, uint32_t floor_cap,
                                 uint32_t *__restrict__ q_cap) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nq) return;
    const uint32_t cnt = (uint32_t)surv_cnt[b];
    const uint32_t cap = cnt > floor_cap ? cnt : floor_cap;
    q_cap[b] = cap > 0xFFFFFF80u ? 0xFFFFFFC0u : ((cap + 63u) & ~63u);
    // (the counter keeps its value: records | runs << 32 of the stage, what the kernels behind the scatter read)
}
// The arena's runs to their queries' segments.  A wave takes 64 runs of a shard at a time: lane r claims its run's place
// with the per-query 64-bit counter (exactly the reservation the direct path makes) and writes the descriptor; the records
// of all 64 runs are then copied by the whole wave, one record per lane and step (a lane finds its run by bisection over
// the chunk's prefix sums, like the replay's fetch), so reads and writes stay as coalesced as the runs are long.
__global__ __launch_bounds__(256) void arena_scatter_kernel(const SurvRec *__restrict__ arena_recs, const uint4 *__restrict__ arena_runs,
                                                            const unsigned long long *__restrict__ arena_cur,
                                                            const unsigned int *__restrict__ arena_fail, uint32_t arena_rsub,
                                                            const unsigned long long *__restrict__ q_base,
                                                            const uint2 *__restrict__ arena_places, SurvRec *__restrict__ surv,
                                                            RunRec *__restrict__ runs) {
    __shared__ uint32_t s_pref[4][65], s_src[4][64];
    __shared__ unsigned long long s_dst[4][64];
    // blocks 0 .. SHARDS-1: the shards (valid runs: up to the first append the shard turned away); block SHARDS: the common area
    // (the common area is walked by RQ_ARENA_COMMON_BLOCKS columns of blocks: it can hold a good part of the runs)
    const bool common = blockIdx.x >= RQ_ARENA_SHARDS;
    const uint32_t shard = common ? RQ_ARENA_SHARDS : blockIdx.x;
    uint32_t nr;
    if (!common) {
        nr = (uint32_t)(arena_cur[shard] >> 32);
        const uint32_t fl = arena_fail[shard];
        nr = nr < fl ? nr : fl;
    } else {
        nr = (uint32_t)(arena_cur[RQ_ARENA_SHARDS + 2] >> 32);
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 *src = arena_runs + (uint64_t)shard * arena_rsub;
    const uint2 *plc = arena_places + (uint64_t)shard * arena_rsub;
    const uint32_t col = common ? blockIdx.x - RQ_ARENA_SHARDS : 0u, ncol = common ? RQ_ARENA_COMMON_BLOCKS : 1u;
    // (a chain of dependent gathers per chunk -- descriptor -> the query's segment start -> the writes; records -> their copies --
    // so the next chunk's descriptors, places and segment starts are requested while this chunk's records move)
    const uint32_t rstep = ncol * gridDim.y * 256, rfirst = ((col * gridDim.y + blockIdx.y) * 4 + wave) * 64;
    uint4 d_n = make_uint4(0, 0, 0, 0);
    uint2 pl_n = make_uint2(0, 0);
    unsigned long long qat_n = 0;
    if (rfirst + lane < nr) {
        d_n = src[rfirst + lane], pl_n = plc[rfirst + lane];
        qat_n = q_base[d_n.z];
    }
    for (uint32_t r0 = rfirst; r0 < nr; r0 += rstep) {
        uint32_t cnt = 0;
        const uint4 d = d_n;
        const uint2 pl = pl_n;  // the run's place inside its query's segment, reserved by the scan
        const unsigned long long qat = qat_n;
        const bool more = r0 + rstep + lane < nr && r0 + rstep >= r0;
        if (more) d_n = src[r0 + rstep + lane], pl_n = plc[r0 + rstep + lane];
        if (r0 + lane < nr) {
            cnt = d.y >> 16;
            const uint32_t base = pl.x, rbase = pl.y;
            RunRec rr;
            rr.pos = d.x, rr.slot = d.y & 0xFFFFu, rr.base = base, rr.cnt = cnt;
            runs[qat + rbase] = rr;
            s_src[wave][lane] = d.w;
            s_dst[wave][lane] = qat + base;
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        s_pref[wave][lane + 1] = incl;
        if (lane == 0) s_pref[wave][0] = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (more) qat_n = q_base[d_n.z];
        const uint32_t total = __shfl(incl, 63, 64);
        for (uint32_t e0 = 0; e0 < total; e0 += 128) {  // two records per lane in flight
            SurvRec rec[2];
            unsigned long long dst[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const uint32_t e = e0 + 64 * v + lane;
                dst[v] = ~0ull;
                if (e < total) {
                    uint32_t lo = 0;  // largest r with s_pref[r] <= e
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1)
                        if (lo + step < 64 && s_pref[wave][lo + step] <= e) lo += step;
                    const uint32_t i = e - s_pref[wave][lo];
                    rec[v] = arena_recs[s_src[wave][lo] + i];
                    dst[v] = s_dst[wave][lo] + i;
                }
            }
#pragma unroll
            for (int v = 0; v < 2; ++v)
                if (dst[v] != ~0ull) surv[dst[v]] = rec[v];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the chunk's LDS rows are free again
    }
}

// exclusive scan of q_cap into q_base (u64); out_total[0] = the sum.  One block of 1024 threads, any nq.
__global__ __launch_bounds__(1024) void seg_scan_kernel(const uint32_t *__restrict__ q_cap, uint32_t nq,
                                                        unsigned long long *__restrict__ q_base, unsigned long long *__restrict__ out_total) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nq; base += 1024) {
        const uint32_t i = base + tid;
        const unsigned long long v = i < nq ? q_cap[i] : 0ull;
        unsigned long long incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += up;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        unsigned long long woff = 0;
        for (uint32_t w = 0; w < wid; ++w) woff += wsum[w];
        const unsigned long long c0 = carry;
        if (i < nq) q_base[i] = c0 + woff + incl - v;
        __syncthreads();
        if (tid == 1023) carry = c0 + woff + incl;
        __syncthreads();
    }
    if (tid == 0) out_total[0] = carry;
}

// dense run directory of a stage: cells [0, ncells) of every query start empty
__global__ void clear_dir_kernel(RunRec *__restrict__ runs, uint32_t nq, const QSeg seg, uint32_t ncells) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nq * ncells) return;
    const uint32_t b = (uint32_t)(i / ncells), c = (uint32_t)(i - (uint64_t)b * ncells);
    RunRec z;
    z.pos = 0, z.slot = 0, z.base = 0, z.cnt = 0;
    runs[seg.at(b) + c] = z;
}

// ranker state of a fresh query (src/rerank.rs:70-77, :129-139) + per-query counters, one launch
// thr_init (seeded passes): the threshold a query starts with instead of f32::MAX, row_map: pass row -> row of thr_init
__global__ void init_state_kernel(ReplayState st, unsigned long long *__restrict__ surv_cnt, uint32_t nq,
                                  const float *__restrict__ thr_init, const uint32_t *__restrict__ row_map) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nq) return;
    st.thr[b] = thr_init ? thr_init[row_map ? row_map[b] : b] : 3.402823466e+38f;  // f32::MAX
    st.recent_max[b] = -3.402823466e+38f;  // f32::MIN
    st.heap_len[b] = 0, st.precise[b] = 0, st.need[b] = 0, st.ovf[b] = 0, st.nsurv[b] = 0, st.nshadow[b] = 0, st.win_count[b] = 0, st.arr_len[b] = 0;
    surv_cnt[b] = 0;
}

// ------------------------------------------------------------------------------------------------
// Results (src/rerank.rs:108-113 heap Vec order; :170-176 the topk smallest, here sorted).
// ------------------------------------------------------------------------------------------------
__global__ void finalize_heap_kernel(const ReplayState st, uint32_t nq, uint32_t topk,
                                     const uint32_t *__restrict__ row_map, const uint32_t *__restrict__ map_ids,
                                     float *__restrict__ out_dist, uint32_t *__restrict__ out_id,
                                     uint32_t *__restrict__ out_n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * topk) return;
    uint32_t b = i / topk, e = i - b * topk;
    uint32_t ob = row_map ? row_map[b] : b;
    uint32_t len = st.heap_len[b];
    if (e < len) {
        out_dist[(uint64_t)ob * topk + e] = ord32_to_f32(st.heap_key[(uint64_t)b * topk + e]);
        out_id[(uint64_t)ob * topk + e] = map_ids[st.heap_id[(uint64_t)b * topk + e]];  // position -> original id
    }
    if (e == 0) out_n[ob] = len;
}

__global__ void finalize_heuristic_kernel(const ReplayState st, uint32_t nq, uint32_t topk,
                                          const uint32_t *__restrict__ row_map, const uint32_t *__restrict__ map_ids,
                                          float *__restrict__ out_dist, uint32_t *__restrict__ out_id,
                                          uint32_t *__restrict__ out_n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq * topk) return;
    uint32_t b = i / topk, e = i - b * topk;
    uint32_t ob = row_map ? row_map[b] : b;
    uint32_t len = st.arr_len[b];
    len = len < st.hcap ? len : st.hcap;
    uint32_t take = len < topk ? len : topk;
    if (e < take) {
        const SurvRec &r = st.arr[(uint64_t)b * st.hcap + e];
        out_dist[(uint64_t)ob * topk + e] = r.rough;
        out_id[(uint64_t)ob * topk + e] = map_ids[__builtin_bit_cast(uint32_t, r.accurate)];  // position -> original id
    }
    if (e == 0) out_n[ob] = take;
}

// per-batch totals for METRICS (src/metrics.rs:44-53): sums of the per-query counters.
// out4[0..5) = {rough, precise (queries without overflow only), #overflowed queries, accurate distances
// computed, max buffer need}
// sums the matrix-core scan's 64 counter pairs into out[0] (sub-tile steps) and out[1] (steps that took the exact path)
__global__ __launch_bounds__(64) void stat_fold_kernel(const unsigned long long *__restrict__ stat, unsigned long long *__restrict__ out) {
    unsigned long long a = stat[2 * threadIdx.x], b = stat[2 * threadIdx.x + 1];
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64), b += __shfl_xor(b, o, 64);
    if (threadIdx.x == 0) out[0] = a, out[1] = b;
}
__global__ __launch_bounds__(256) void metrics_sum_kernel(const unsigned long long *__restrict__ rough,
                                                          const uint32_t *__restrict__ precise,
                                                          const uint32_t *__restrict__ need,
                                                          const uint32_t *__restrict__ arr_len,
                                                          const uint32_t *__restrict__ nsurv,
                                                          const uint32_t *__restrict__ nshadow, uint32_t nq,
                                                          const uint32_t *__restrict__ ovf, uint32_t hcap,
                                                          unsigned long long *__restrict__ out4) {
    __shared__ unsigned long long s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long r = 0, p = 0, o = 0, a = 0, mx = 0, sh = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
        const bool ok = !ovf[i] && (!arr_len || arr_len[i] <= hcap);
        r += rough[i];
        p += ok ? precise[i] : 0;
        o += ok ? 0 : 1;
        a += nsurv[i];
        sh += nshadow[i];
        const unsigned long long al = arr_len ? arr_len[i] : 0ull;
        unsigned long long nd = need[i] > al ? (unsigned long long)need[i] : al;
        mx = nd > mx ? nd : mx;
    }
    atomicAdd(&s[0], r);
    atomicAdd(&s[1], p);
    atomicAdd(&s[2], o);
    atomicAdd(&s[3], a);
    atomicMax(&s[4], mx);
    atomicAdd(&s[5], sh);
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&out4[threadIdx.x], s[threadIdx.x]);
    if (threadIdx.x == 5) atomicAdd(&out4[5], s[5]);
    if (threadIdx.x == 4) atomicMax(&out4[4], s[4]);
}
