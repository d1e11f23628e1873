// kernels_build.h -- gfx950 kernels for RaBitQ::from_path (src/rabitq.rs:159-265).
#pragma once
#include "common.h"
#include "kernels_query.h"

#pragma clang fp contract(off)

#ifndef RQ_F32X16_DEFINED
#define RQ_F32X16_DEFINED
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

// ------------------------------------------------------------------------------------------------
// Rotation X' = X P on the matrix cores (src/rabitq.rs:188-189; query side src/utils.rs:237-258).
//
// Bit-exact with the reference's AVX2 `vector_dot_product` order (src/simd.rs:257-314) BY
// CONSTRUCTION: that routine keeps 8 independent accumulators, lane l running the fused chain
//     acc_l = fma(x[8c+l], P[8c+l][j], acc_l)   for c = 0, 1, ...
// and folds them as ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)).  v_mfma_f32_32x32x2_f32 computes, per
// output element, D = fma(A[i][1], B[1][j], fma(A[i][0], B[0][j], C)) -- a k-ordered f32 fmaf
// chain with one rounding per product.  So each AVX lane l gets its OWN 32x32 accumulator tile,
// fed with k = 8c + l for consecutive chunk pairs (c, c+1); eight tiles (128 accumulator VGPRs)
// are folded in the AVX order in the epilogue.  Same flops as a plain GEMM (2*dim per output).
//
// block = 256 threads = 4 waves as 2(M) x 2(N); block tile 64 rows x 64 cols; K streamed through
// LDS in slabs of 64 (X slab padded to 65 floats per row: conflict-free ds_read_b32 down a column).
// ------------------------------------------------------------------------------------------------
#define ROT_BM 64
#define ROT_BN 64
#define ROT_BK 64

struct RotStage {  // one pipeline item's global->register staging: 64x64 X slab + 64x64 P slab
    float4 x[4], p[4];
};

__device__ __forceinline__ void rot_load(RotStage &st, const float *__restrict__ x, const float *__restrict__ P,
                                         uint64_t n, uint32_t dim, uint64_t row0, uint32_t col0, uint32_t k0,
                                         uint32_t tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        uint32_t idx = tid + 256 * it;  // 0..1023 float4 slots
        uint32_t rr = idx >> 4, c4 = (idx & 15) * 4;
        st.x[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + rr < n) st.x[it] = *reinterpret_cast<const float4 *>(x + (row0 + rr) * dim + k0 + c4);
        st.p[it] = *reinterpret_cast<const float4 *>(P + (uint64_t)(k0 + rr) * dim + col0 + c4);
    }
}

__device__ __forceinline__ void rot_store_lds(const RotStage &st, float (*Xs)[ROT_BK + 1], float (*Ps)[ROT_BN],
                                              uint32_t tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        uint32_t idx = tid + 256 * it;
        uint32_t rr = idx >> 4, c4 = (idx & 15) * 4;
        Xs[rr][c4] = st.x[it].x, Xs[rr][c4 + 1] = st.x[it].y, Xs[rr][c4 + 2] = st.x[it].z, Xs[rr][c4 + 3] = st.x[it].w;
        *reinterpret_cast<float4 *>(&Ps[rr][c4]) = st.p[it];
    }
}

template <bool FIRST>
__device__ __forceinline__ void rot_compute(f32x16 (&acc)[8], const float (*Xs)[ROT_BK + 1],
                                            const float (*Ps)[ROT_BN], uint32_t wr, uint32_t wc, uint32_t li,
                                            uint32_t kk) {
#pragma unroll
    for (int cp = 0; cp < ROT_BK / 16; ++cp) {
        const uint32_t kbase = 8 * (2 * cp + kk);  // this half-wave's chunk inside the slab
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float a = Xs[wr * 32 + li][kbase + l];
            float b = Ps[kbase + l][wc * 32 + li];
            if (FIRST && cp == 0) {
                f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // C = inline constant 0
                acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, zero, 0, 0, 0);
            } else {
                acc[l] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[l], 0, 0, 0);
            }
        }
    }
}

// Persistent: block (col tile c, group g) walks row tiles g, g+G, ... ; the (row tile, K slab) items
// form one software-pipelined stream: while item i is on the matrix cores out of LDS buffer i&1, the
// global loads of item i+1 are in flight into registers and are written to buffer (i+1)&1 after the
// MFMAs; one barrier per item.
__global__ __launch_bounds__(256, 2) void rotate_mfma_kernel(const float *__restrict__ x,
                                                             const float *__restrict__ P,
                                                             float *__restrict__ out, uint64_t n,
                                                             uint32_t dim, uint32_t groups) {
    __shared__ float Xs[2][ROT_BM][ROT_BK + 1];
    __shared__ __attribute__((aligned(16))) float Ps[2][ROT_BK][ROT_BN];
    const uint32_t ncol_tiles = dim / ROT_BN;
    const uint32_t col_tile = blockIdx.x % ncol_tiles, g = blockIdx.x / ncol_tiles;
    const uint32_t col0 = col_tile * ROT_BN;
    const uint64_t nrow_tiles = (n + ROT_BM - 1) / ROT_BM;
    if (g >= nrow_tiles) return;
    const uint64_t my_tiles = (nrow_tiles - g + groups - 1) / groups;
    const uint32_t nslab = dim / ROT_BK;
    const uint64_t total = my_tiles * nslab;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wr = wave >> 1, wc = wave & 1;
    const uint32_t li = lane & 31, kk = lane >> 5;

    f32x16 acc[8];
    RotStage st;
    rot_load(st, x, P, n, dim, (uint64_t)g * ROT_BM, col0, 0, tid);
    rot_store_lds(st, Xs[0], Ps[0], tid);
    __syncthreads();
    uint64_t tile = g;
    uint32_t slab = 0;
    for (uint64_t it = 0; it < total; ++it) {
        // next item's coordinates
        uint64_t ntile = tile;
        uint32_t nsl = slab + 1;
        if (nsl == nslab) {
            nsl = 0;
            ntile = tile + groups;
        }
        const bool more = it + 1 < total;
        if (more) rot_load(st, x, P, n, dim, ntile * ROT_BM, col0, nsl * ROT_BK, tid);
        const uint32_t buf = (uint32_t)(it & 1);
        if (slab == 0) rot_compute<true>(acc, Xs[buf], Ps[buf], wr, wc, li, kk);
        else rot_compute<false>(acc, Xs[buf], Ps[buf], wr, wc, li, kk);
        if (slab + 1 == nslab) {
            // epilogue: AVX fold per element; C/D map: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            const uint64_t row0 = tile * ROT_BM;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float c0 = acc[0][r] + acc[4][r], c1 = acc[1][r] + acc[5][r];
                float c2 = acc[2][r] + acc[6][r], c3 = acc[3][r] + acc[7][r];
                float v = (c0 + c1) + (c2 + c3);
                uint64_t row = row0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row < n) out[row * dim + col0 + wc * 32 + li] = v;
            }
        }
        if (more) rot_store_lds(st, Xs[buf ^ 1], Ps[buf ^ 1], tid);
        __syncthreads();
        tile = ntile;
        slab = nsl;
    }
}

// transpose k x dim -> dim x k (rotated centroids for the coalesced lane<->centroid kernels)
__global__ void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, uint32_t rows,
                                 uint32_t cols) {
    __shared__ float t[32][33];
    uint32_t c = blockIdx.x * 32 + threadIdx.x, r = blockIdx.y * 32 + threadIdx.y;
    for (int i = 0; i < 32; i += 8)
        if (r + i < rows && c < cols) t[threadIdx.y + i][threadIdx.x] = in[(uint64_t)(r + i) * cols + c];
    __syncthreads();
    uint32_t oc = blockIdx.y * 32 + threadIdx.x, orow = blockIdx.x * 32 + threadIdx.y;
    for (int i = 0; i < 32; i += 8)
        if (orow + i < cols && oc < rows) out[(uint64_t)(orow + i) * rows + oc] = t[threadIdx.x][threadIdx.y + i];
}

// ------------------------------------------------------------------------------------------------
// kmeans_nearest_cluster (src/utils.rs:261-277): label = first argmin_j l2_squared_distance(c'_j, x')
// with strict `<` from (0, f32::MAX), every distance in the exact order of src/simd.rs:14-73.
//
// Fast path, dim <= 128: lane <-> vector with the whole rotated vector in VGPRs; centroids are
// streamed through LDS in tiles and read as wave-uniform broadcasts, so the inner loop is pure
// VALU (v_sub + v_fma per element) and each lane sees centroids in ascending j, which gives the
// reference's first-minimum tie-break with no cross-lane traffic at all.
// ------------------------------------------------------------------------------------------------
#define ASSIGN_CT 32
template <int DIM>
__global__ __launch_bounds__(256) void assign_regs_kernel(const float *__restrict__ xrot,
                                                          const float *__restrict__ centroids, uint64_t n,
                                                          uint32_t k, uint32_t *__restrict__ label,
                                                          float *__restrict__ dist) {
    __shared__ __attribute__((aligned(16))) float cs[ASSIGN_CT][DIM];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    float xv[DIM];
    {
        const float4 *src = reinterpret_cast<const float4 *>(xrot + (live ? i : 0) * DIM);
#pragma unroll
        for (int e = 0; e < DIM / 4; ++e) {
            float4 v = src[e];
            xv[4 * e] = v.x, xv[4 * e + 1] = v.y, xv[4 * e + 2] = v.z, xv[4 * e + 3] = v.w;
        }
    }
    float best = 3.402823466e+38f;
    uint32_t lab = 0;
    for (uint32_t j0 = 0; j0 < k; j0 += ASSIGN_CT) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < ASSIGN_CT * DIM / 4; t += 256) {
            uint32_t row = t / (DIM / 4), c4 = t - row * (DIM / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j0 + row < k) v = reinterpret_cast<const float4 *>(centroids + (uint64_t)(j0 + row) * DIM)[c4];
            reinterpret_cast<float4 *>(&cs[row][0])[c4] = v;
        }
        __syncthreads();
        const uint32_t lim = (k - j0) < ASSIGN_CT ? (k - j0) : ASSIGN_CT;
        for (uint32_t ct = 0; ct < lim; ++ct) {
            // the 8 AVX-lane chains as 4 packed pairs: v_pk_add_f32 (diff) + v_pk_fma_f32, each component
            // rounded exactly like the scalar sub / fma
            f32x2 acc2[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
            for (int c = 0; c < DIM; c += 8) {
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    f32x2 cv = {cs[ct][c + 2 * l], cs[ct][c + 2 * l + 1]};
                    f32x2 xx = {xv[c + 2 * l], xv[c + 2 * l + 1]};
                    f32x2 d = cv - xx;
                    acc2[l] = __builtin_elementwise_fma(d, d, acc2[l]);
                }
            }
            float acc[8] = {acc2[0].x, acc2[0].y, acc2[1].x, acc2[1].y, acc2[2].x, acc2[2].y, acc2[3].x, acc2[3].y};
            float dd = reduce8_regs(acc);
            if (dd < best) {
                best = dd;
                lab = j0 + ct;
            }
        }
    }
    if (live) {
        label[i] = lab;
        dist[i] = best;
    }
}

// Generic path (any dim): lane <-> centroid over the transposed centroids, VT vectors per block in
// LDS, block-wide argmin on the unique composite key (biased Ord32(dist) << 32 | j): the minimum
// of that key is the first minimum.
template <int VT>
__global__ __launch_bounds__(256) void assign_generic_kernel(const float *__restrict__ xrot,
                                                             const float *__restrict__ cent_t, uint64_t n,
                                                             uint32_t k, uint32_t dim,
                                                             uint32_t *__restrict__ label,
                                                             float *__restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [dim][VT]: one ds_read_b128 = 4 vectors at one dimension
    __shared__ unsigned long long red[VT][4];
    static_assert(VT % 4 == 0, "vectors are processed in packed pairs, read four at a time");
    const uint64_t i0 = (uint64_t)blockIdx.x * VT;
    for (uint32_t t = threadIdx.x; t < VT * dim; t += 256) {
        uint32_t v = t / dim, e = t - v * dim;  // coalesced reads of the rows, transposed into LDS
        xs[e * VT + v] = (i0 + v < n) ? xrot[(i0 + v) * dim + e] : 0.0f;
    }
    __syncthreads();
    unsigned long long bestkey[VT];
#pragma unroll
    for (int v = 0; v < VT; ++v) bestkey[v] = ((unsigned long long)ord32_biased(3.402823466e+38f) << 32);
    for (uint32_t j = threadIdx.x; j < k; j += 256) {
        f32x2 acc2[VT / 2][8];  // [vector pair][AVX lane]: v_pk_add_f32 + v_pk_fma_f32, per-component rounding
#pragma unroll
        for (int v = 0; v < VT / 2; ++v)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc2[v][l] = f32x2{0.0f, 0.0f};
        const float *cp = cent_t + j;
        for (uint32_t c = 0; c < dim; c += 8) {
            float ce[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) ce[l] = cp[(uint64_t)(c + l) * k];  // 8 loads in flight
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const f32x2 ce2 = {ce[l], ce[l]};
#pragma unroll
                for (int v4 = 0; v4 < VT / 4; ++v4) {
                    const float4 xq = *reinterpret_cast<const float4 *>(&xs[(c + l) * VT + 4 * v4]);
                    const f32x2 x01 = {xq.x, xq.y}, x23 = {xq.z, xq.w};
                    const f32x2 d01 = ce2 - x01, d23 = ce2 - x23;  // centroid - vector, as l2_squared_distance(c', x') does
                    acc2[2 * v4][l] = __builtin_elementwise_fma(d01, d01, acc2[2 * v4][l]);
                    acc2[2 * v4 + 1][l] = __builtin_elementwise_fma(d23, d23, acc2[2 * v4 + 1][l]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VT; ++v) {
            float accv[8];
#pragma unroll
            for (int l = 0; l < 8; ++l) accv[l] = (v & 1) ? acc2[v / 2][l].y : acc2[v / 2][l].x;
            float dd = reduce8_regs(accv);
            // strict `<` against f32::MAX start: NaN and >= MAX never win (utils.rs:271)
            if (dd < 3.402823466e+38f) {
                unsigned long long key = ((unsigned long long)ord32_biased(dd) << 32) | j;
                bestkey[v] = key < bestkey[v] ? key : bestkey[v];
            }
        }
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < VT; ++v) {
        unsigned long long key = bestkey[v];
        for (int o = 32; o >= 1; o >>= 1) {
            unsigned long long other = __shfl_xor(key, o, 64);
            key = other < key ? other : key;
        }
        if (lane == 0) red[v][wave] = key;
    }
    __syncthreads();
    if (threadIdx.x < VT && i0 + threadIdx.x < n) {
        unsigned long long key = red[threadIdx.x][0];
        for (int w = 1; w < 4; ++w) key = red[threadIdx.x][w] < key ? red[threadIdx.x][w] : key;
        label[i0 + threadIdx.x] = (uint32_t)key;
        dist[i0 + threadIdx.x] = ord32_unbias((uint32_t)(key >> 32));
    }
}

// ------------------------------------------------------------------------------------------------
// Residual sign-pack + factors (src/rabitq.rs:205-229, src/utils.rs:53-67).  8 lanes per vector,
// lane l <-> AVX lane l (elements 8c + l):
//   sq   = l2_squared_distance(x', c')           (faer norm_l2 at :206 -> sqrtf of the AVX-ordered sum)
//   code bit e = (x'_e - c'_e > 0)               (strict, utils.rs:56)
//   num  = <r, sign(r)> in vector_dot_product order (faer inner product at :212)
//   ip   = num / (norm*sqrt(dim)) if that norm is_normal else 0.8          (:210-215)
//   error_bound = (2*1.9/sqrt(dim-1)) * sqrt((norm/ip)^2 - norm^2), factor_ip = -2/sqrt(dim) * norm/ip,
//   factor_ppc = factor_ip * (2*popcount - dim)                            (:220-229)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_kernel(const float *__restrict__ xrot,
                                                       const float *__restrict__ centroids,
                                                       const uint32_t *__restrict__ label, uint64_t n,
                                                       uint32_t dim, uint64_t *__restrict__ codes,
                                                       float4 *__restrict__ factors) {
    const uint32_t l = threadIdx.x & 7;
    const uint64_t i = (uint64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (i >= n) return;
    const float *x = xrot + i * dim;
    const float *c = centroids + (uint64_t)label[i] * dim;
    const uint32_t W = dim >> 6;
    float acc_sq = 0.0f, acc_dot = 0.0f;
    uint32_t pop = 0;
    for (uint32_t w = 0; w < W; ++w) {
        uint32_t lo32 = 0, hi32 = 0;
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
            uint32_t e = 64 * w + 8 * cc + l;
            float xe = x[e], ce = c[e];
            float r = xe - ce;  // rabitq.rs:205
            acc_sq = fmaf(r, r, acc_sq);
            bool posv = r > 0.0f;
            float sg = posv ? 1.0f : -1.0f;
            acc_dot = fmaf(r, sg, acc_dot);
            uint32_t bit = posv ? 1u : 0u;
            pop += bit;
            if (cc < 4) lo32 |= bit << (8 * cc + l);
            else hi32 |= bit << (8 * (cc - 4) + l);
        }
        for (int o = 1; o < 8; o <<= 1) {
            lo32 |= __shfl_xor(lo32, o, 8);
            hi32 |= __shfl_xor(hi32, o, 8);
        }
        if (l == 0) codes[i * W + w] = ((uint64_t)hi32 << 32) | lo32;
    }
    float sq = reduce8_lanes(acc_sq);
    float num = reduce8_lanes(acc_dot);
    for (int o = 1; o < 8; o <<= 1) pop += __shfl_xor(pop, o, 8);
    if (l == 0) {
        const float dim_sqrt = sqrtf((float)dim);
        float norm = sqrtf(sq);
        float cds = norm * norm;
        float nd = norm * dim_sqrt;
        float and_ = fabsf(nd);
        bool normal = (and_ >= 1.17549435e-38f) && (and_ <= 3.402823466e+38f);  // f32::is_normal
        float ip = normal ? num / nd : 0.8f;
        float over = norm / ip;
        const float error_base = 2.0f * 1.9f / sqrtf((float)dim - 1.0f);
        float4 f;
        f.z = error_base * sqrtf(over * over - cds);     // error_bound
        f.x = -2.0f / dim_sqrt * over;                   // factor_ip
        f.y = f.x * (float)(2 * (int)pop - (int)dim);    // factor_ppc
        f.w = cds;                                       // center_distance_square
        factors[i] = f;
    }
}

// ------------------------------------------------------------------------------------------------
// Cluster ordering (src/rabitq.rs:232-252): per-list stable sort by centroid distance (ties keep
// ascending original id), offsets = prefix sum, gather base/codes/factors.  The composite key
// (biased Ord32(dist) << 32 | id) is unique, so bucketing with atomics followed by an ordinary
// per-list sort of that key yields exactly the stable order.
// ------------------------------------------------------------------------------------------------
__global__ void label_hist_kernel(const uint32_t *__restrict__ label, uint64_t n, uint32_t *__restrict__ cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[label[i]], 1u);
}

__global__ void label_scatter_kernel(const uint32_t *__restrict__ label, const float *__restrict__ dist,
                                     uint64_t n, uint64_t id0, const uint32_t *__restrict__ offsets,
                                     uint32_t *__restrict__ cursor, unsigned long long *__restrict__ keys) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t c = label[i];
    uint32_t at = atomicAdd(&cursor[c], 1u);
    keys[(uint64_t)offsets[c] + at] = ((unsigned long long)ord32_biased(dist[i]) << 32) | (uint32_t)(id0 + i);
}

#define RQ_LISTSORT_LDS 8192
__global__ __launch_bounds__(1024) void list_sort_kernel(unsigned long long *__restrict__ keys,
                                                         const uint32_t *__restrict__ offsets) {
    __shared__ unsigned long long lds[RQ_LISTSORT_LDS];
    const uint32_t c = blockIdx.x;
    const uint32_t b = offsets[c], n = offsets[c + 1] - b;
    if (n < 2) return;
    unsigned long long *seg = keys + b;
    auto key = [](unsigned long long v) { return v; };
    if (n <= RQ_LISTSORT_LDS) {
        for (uint32_t i = threadIdx.x; i < n; i += 1024) lds[i] = seg[i];
        __syncthreads();
        bitonic_sort_block(lds, n, key);
        for (uint32_t i = threadIdx.x; i < n; i += 1024) seg[i] = lds[i];
    } else {
        bitonic_sort_block(seg, n, key);
    }
}

// Cluster order, small arrays: destination position p takes the code / factors of original vector id = low 32 bits of
// the sorted key; map_ids[p] = id (src/rabitq.rs:250) and the inverse permutation pos_of_id[id] = p, which the
// placement pass uses to put every raw vector at its final position.  Grid-stride (HIP silently wraps a launch
// whose gridDim.x * blockDim.x reaches 2^32).
__global__ __launch_bounds__(256) void order_gather_kernel(const unsigned long long *__restrict__ keys, uint64_t n, uint32_t W,
                                                           const uint64_t *__restrict__ codes_in,
                                                           const float4 *__restrict__ factors_in,
                                                           uint64_t *__restrict__ codes_out, float4 *__restrict__ factors_out,
                                                           uint32_t *__restrict__ map_ids, uint32_t *__restrict__ pos_of_id) {
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (uint64_t)gridDim.x * 256) {
        const uint32_t id = (uint32_t)keys[p];
        for (uint32_t w = 0; w < W; ++w) codes_out[p * W + w] = codes_in[(uint64_t)id * W + w];
        factors_out[p] = factors_in[id];
        map_ids[p] = id;
        pos_of_id[id] = (uint32_t)p;
    }
}

// Cluster order, raw vectors (src/rabitq.rs:244-247): rows i0 .. i0+m of the input (m x d, un-rotated) go to their
// final positions, zero-padded to dim, in whichever tier holds that position (HBM, or pinned host memory written
// over the host link).  One wave per row: contiguous read, contiguous 4*dim-byte write.
__global__ __launch_bounds__(256) void place_rows_kernel(const float *__restrict__ rows, uint64_t i0, uint64_t m, uint32_t d,
                                                         uint32_t dim, const uint32_t *__restrict__ pos_of_id,
                                                         const BaseView out) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < m; r += (uint64_t)gridDim.x * 4) {
        const RowRef dst = out.row(pos_of_id[i0 + r], dim);
        const float *src = rows + r * d;
        if (!dst.split) {
            float *w = const_cast<float *>(dst.p);
            for (uint32_t e = lane; e < dim; e += 64) w[e] = e < d ? src[e] : 0.0f;
        } else {  // two elements per lane: one 4-byte store into each plane (dim is a multiple of 64)
            uint32_t *hp = reinterpret_cast<uint32_t *>(const_cast<float *>(dst.p)), *lp = hp + dim / 2;
            for (uint32_t e = 2 * lane; e < dim; e += 128) {
                const uint32_t b0 = __builtin_bit_cast(uint32_t, e < d ? src[e] : 0.0f);
                const uint32_t b1 = __builtin_bit_cast(uint32_t, e + 1 < d ? src[e + 1] : 0.0f);
                hp[e / 2] = ((b0 + 0x8000u) >> 16) | ((b1 + 0x8000u) & 0xFFFF0000u);
                lp[e / 2] = (b0 & 0xFFFFu) | (b1 << 16);
            }
        }
    }
}

// out[0] = longest list, out[1] = shortest list (0 if any list is empty); out preset to {0, 0xFFFFFFFF}
__global__ void max_list_len_kernel(const uint32_t *__restrict__ offsets, uint32_t k, uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) {
        atomicMax(out, offsets[i + 1] - offsets[i]);
        atomicMin(out + 1, offsets[i + 1] - offsets[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// Centroid training (the crate takes centroids as an input; scripts/cluster.py:63-108 trains them
// with faiss k-means on a 256-points-per-centroid sample).  Lloyd iterations on a sample, reusing
// the nearest-centroid kernels above.  No parity target (faiss is absent, results are seed
// dependent): judged by recall only.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rq_mix64(uint64_t x) {  // splitmix64
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void kmeans_sample_kernel(const float *__restrict__ x, uint64_t n, uint32_t d, uint32_t dim,
                                     uint64_t seed, uint64_t ns, float *__restrict__ xs) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns * dim) return;
    uint64_t r = i / dim;
    uint32_t c = (uint32_t)(i - r * dim);
    uint64_t src = rq_mix64(seed * 0x100000001B3ull + r) % n;
    xs[i] = c < d ? x[src * d + c] : 0.0f;
}

__global__ void kmeans_accumulate_kernel(const float *__restrict__ xs, const uint32_t *__restrict__ label,
                                         uint64_t ns, uint32_t dim, float *__restrict__ sums,
                                         uint32_t *__restrict__ counts) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns * dim) return;
    uint64_t r = i / dim;
    uint32_t c = (uint32_t)(i - r * dim);
    uint32_t l = label[r];
    atomicAdd(&sums[(uint64_t)l * dim + c], xs[i]);
    if (c == 0) atomicAdd(&counts[l], 1u);
}

__global__ void kmeans_update_kernel(const float *__restrict__ sums, const uint32_t *__restrict__ counts,
                                     const float *__restrict__ xs, uint64_t ns, uint32_t k, uint32_t dim,
                                     uint64_t seed, float *__restrict__ centroids) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)k * dim) return;
    uint32_t j = (uint32_t)(i / dim), c = (uint32_t)(i - (uint64_t)j * dim);
    uint32_t cnt = counts[j];
    if (cnt) centroids[i] = sums[i] / (float)cnt;
    else centroids[i] = xs[(rq_mix64(seed ^ (0xABCDull + j)) % ns) * dim + c];  // re-seed an empty cluster
}

__global__ void unpad_rows_kernel(const float *__restrict__ in, float *__restrict__ out, uint64_t n, uint32_t dim,
                                  uint32_t d) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * d) return;
    uint64_t r = i / d;
    out[i] = in[r * dim + (i - r * d)];
}

// index-wide bounds of the Factor fields (see FactorStats); non-finite / irregular entries are skipped:
// such candidates take the exact path in the matrix-core scan anyway
__global__ __launch_bounds__(256) void factor_stats_kernel(const float4 *__restrict__ factors, uint64_t n,
                                                           uint32_t *__restrict__ out4 /* float bits, >= 0 */) {
    float cds = 0.0f, ppc = 0.0f, eb = 0.0f, inv = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const float4 f = factors[i];
        const float r = 1.0f / f.x;
        const float mag = fabsf(r) + fabsf(f.w) + fabsf(f.y) + fabsf(f.z);
        if (f.x < 0.0f && mag < 3.0e38f) {
            cds = fmaxf(cds, fabsf(f.w)), ppc = fmaxf(ppc, fabsf(f.y)), eb = fmaxf(eb, fabsf(f.z)), inv = fmaxf(inv, fabsf(r));
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        cds = fmaxf(cds, __shfl_xor(cds, o, 64)), ppc = fmaxf(ppc, __shfl_xor(ppc, o, 64));
        eb = fmaxf(eb, __shfl_xor(eb, o, 64)), inv = fmaxf(inv, __shfl_xor(inv, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {  // non-negative floats order like their bit patterns
        atomicMax(&out4[0], __builtin_bit_cast(uint32_t, cds));
        atomicMax(&out4[1], __builtin_bit_cast(uint32_t, ppc));
        atomicMax(&out4[2], __builtin_bit_cast(uint32_t, eb));
        atomicMax(&out4[3], __builtin_bit_cast(uint32_t, inv));
    }
}

// Per-list reference U0 of u' = (1, cds, ppc, eb) / factor_ip for the additive gate of the matrix-core scan (kernels_query.h,
// scan_mfma_kernel<.., ADD>): the mean over the list's regular vectors of components 0, 1 and 3 (component 2 = 2 popcount - dim is
// centred on 0 by construction and keeps the reference 0).  ANY value is a valid reference -- the bound it enters holds for every
// U0 -- a good one only keeps |u' - U0| small.  One block per list.
__global__ __launch_bounds__(256) void list_uref_kernel(const float4 *__restrict__ factors, const uint32_t *__restrict__ offsets,
                                                        float4 *__restrict__ uref) {
    const uint32_t c = blockIdx.x, b = offsets[c], e = offsets[c + 1];
    float s0 = 0.0f, s1 = 0.0f, s3 = 0.0f, n = 0.0f;
    for (uint32_t i = b + threadIdx.x; i < e; i += 256) {
        const float4 f = factors[i];
        const float rf = __builtin_amdgcn_rcpf(f.x);
        const float mag = (1.0f + fabsf(f.w) + fabsf(f.y) + fabsf(f.z)) * fabsf(rf);
        if (f.x < 0.0f && mag < 1.0e37f) s0 += rf, s1 += f.w * rf, s3 += f.z * rf, n += 1.0f;
    }
    for (int o = 32; o >= 1; o >>= 1)
        s0 += __shfl_xor(s0, o, 64), s1 += __shfl_xor(s1, o, 64), s3 += __shfl_xor(s3, o, 64), n += __shfl_xor(n, o, 64);
    __shared__ float red[4][4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][0] = s0, red[threadIdx.x >> 6][1] = s1, red[threadIdx.x >> 6][2] = s3, red[threadIdx.x >> 6][3] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[4];
        for (int r = 0; r < 4; ++r) t[r] = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
        const float inv = t[3] > 0.0f ? 1.0f / t[3] : 0.0f;
        float4 u = make_float4(t[0] * inv, t[1] * inv, 0.0f, t[2] * inv);
        if (!(fabsf(u.x) + fabsf(u.y) + fabsf(u.w) < 1.0e37f)) u = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uref[c] = u;
    }
}

// ------------------------------------------------------------------------------------------------
// Nearest list through the matrix cores, WITHOUT giving up the exact result (kmeans_nearest_cluster,
// src/utils.rs:261-277: the first j minimising the exact-order f32 distance).
//
// The exact-order VALU kernels above are 3 n k dim flop of vector work: 2.9 of the 3.1 s of a 100M x 128 build.
// Here every (vector, centroid) distance is first APPROXIMATED as (|c|^2 + |x|^2) - 2 <c~, x~> with the inner products
// from v_mfma_f32_32x32x16_bf16 (operands rounded to bf16, f32 accumulation: 16x the f32 MFMA rate), which is within
//     m_x = (2^-8 + (2 dim + 64) 2^-24) 1.05 (Cmax + |x|)^2
// of the reference's f32 value e_j (src/simd.rs:14-73).  bf16 keeps 8 significant bits: round-to-nearest is 2^-8 relative per
// operand, so |<x,c> - <x~,c~>| <= (2^-7 + 2^-16) |x||c|, the term -2<x,c> of the distance is off by at most
// (2^-6 + 2^-15) |x||c| <= (2^-8 + 2^-17) (|x|+|c|)^2   (|x||c| <= (|x|+|c|)^2 / 4): the first term of m_x, with its 2^-17
// tail inside the factor 1.05 -- that factor is LOAD-BEARING (it is the only slack over the bf16 bound: do not tighten it).
// The f32 accumulation of the 128-term products, the two norms (a dim-long f32 fma chain each) and the reference's own chain
// ((dim/8 + 5) 2^-24 relative) are all inside the second term; Cmax = the largest centroid norm.  So the exact minimiser j*
// (and every exact tie) has
//     a_{j*} <= e_{j*} + m <= e_j + m <= a_j + 2 m   for every j,   in particular   a_{j*} <= a_min + 2 m:
// pass 1 finds a_min per vector, pass 2 lists the lists with a <= a_min + 2 m (one on well separated data, a few on
// overlapping clusters), assign_refine_kernel recomputes THOSE in the reference's lane order and takes the first
// minimum.  Vectors with no candidate (non-finite input) or more than RQ_ASSIGN_CAND of them (near-equidistant
// centroids) are listed for the exact-order kernel.  Labels and distances: bit-identical to the kernels above.
//
// Geometry: block = 4 waves; a wave keeps the bf16 MFMA B-fragments of NT x 32 vectors in registers for the whole
// kernel (dim/16 fragments of 4 VGPRs per tile) and streams all centroid tiles (32 lists x dim bf16, pre-rounded once
// per build) through a double-buffered LDS image (row stride dim * 2 + 16 bytes).  D lane map: column = vector
// (lane & 31), the 16 registers x 2 half-waves = the 32 lists of the tile.
// ------------------------------------------------------------------------------------------------
#define RQ_ASSIGN_CAND 16u
typedef __bf16 asg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float asg_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t asg_bf16_pair(float lo, float hi) {  // two f32 -> packed bf16 (round to nearest even)
    uint32_t a = __builtin_bit_cast(uint32_t, lo), b = __builtin_bit_cast(uint32_t, hi);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}
// rows x dim f32 -> bf16 (round to nearest even), 8 elements per thread
__global__ __launch_bounds__(256) void to_bf16_kernel(const float *__restrict__ in, uint64_t total, uint16_t *__restrict__ out) {
    const uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= total) return;
    const float4 a = *reinterpret_cast<const float4 *>(in + i), b = *reinterpret_cast<const float4 *>(in + i + 4);
    *reinterpret_cast<uint4 *>(out + i) = make_uint4(asg_bf16_pair(a.x, a.y), asg_bf16_pair(a.z, a.w), asg_bf16_pair(b.x, b.y), asg_bf16_pair(b.z, b.w));
}
// squared norms of n rows (sequential f32 fma chain per row) and their maximum (as u32 bits: non-negative floats order like
// their bit patterns; NaN / inf patterns sort above every finite norm)
__global__ void row_sqnorm_kernel(const float *__restrict__ rows, uint32_t n, uint32_t dim, float *__restrict__ out,
                                  uint32_t *__restrict__ max_bits) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float4 *p = reinterpret_cast<const float4 *>(rows + (uint64_t)r * dim);
    float s = 0.0f;
    for (uint32_t c = 0; c < dim / 4; ++c) {
        const float4 v = p[c];
        s = fmaf(v.x, v.x, s), s = fmaf(v.y, v.y, s), s = fmaf(v.z, v.z, s), s = fmaf(v.w, v.w, s);
    }
    out[r] = s;
    if (max_bits) atomicMax(max_bits, __builtin_bit_cast(uint32_t, s));
}

template <int W, int NT>
__global__ __launch_bounds__(256, (W <= 4 ? 2 : 1)) void assign_approx_kernel(const float *__restrict__ xrot,
                                                              const uint16_t *__restrict__ cent_bf /* k x dim bf16 */,
                                                              const float *__restrict__ cnorm, float cmax, uint64_t n,
                                                              uint32_t k, uint32_t *__restrict__ cand /* n x RQ_ASSIGN_CAND */,
                                                              uint32_t *__restrict__ cand_cnt /* n, zeroed */) {
    constexpr int DIM = 64 * W, NM = DIM / 16;             // MFMAs per (vector tile, centroid tile)
    constexpr uint32_t ROWB = DIM * 2 + 16;                // LDS row stride of a centroid tile, bytes
    constexpr uint32_t TILEB = 32 * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char asg_lds[];  // 2 x (tile image | 32 norms)
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 31, kh = lane >> 5;
    const uint64_t v0 = ((uint64_t)blockIdx.x * 4 + wave) * (32 * NT);
    // ---- this wave's vectors: bf16 B fragments (k-elements 16 m + 8 kh .. + 7 of vector v0 + 32 tile + col) and |x|^2 -------
    asg_bf16x8 bfrag[NT][NM];
    float xn[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const uint64_t v = v0 + 32 * tl + col;
        const float *xp = xrot + (v < n ? v : (n - 1)) * DIM + 8 * kh;
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const float4 a = *reinterpret_cast<const float4 *>(xp + 16 * m), b = *reinterpret_cast<const float4 *>(xp + 16 * m + 4);
            s0 = fmaf(a.x, a.x, s0), s1 = fmaf(a.y, a.y, s1), s0 = fmaf(a.z, a.z, s0), s1 = fmaf(a.w, a.w, s1);
            s0 = fmaf(b.x, b.x, s0), s1 = fmaf(b.y, b.y, s1), s0 = fmaf(b.z, b.z, s0), s1 = fmaf(b.w, b.w, s1);
            const uint4 pk = make_uint4(asg_bf16_pair(a.x, a.y), asg_bf16_pair(a.z, a.w), asg_bf16_pair(b.x, b.y), asg_bf16_pair(b.z, b.w));
            bfrag[tl][m] = __builtin_bit_cast(asg_bf16x8, pk);
        }
        const float s = s0 + s1;
        xn[tl] = s + __shfl_xor(s, 32, 64);  // both halves of the vector
    }
    const uint32_t ntile = (k + 31) / 32;
    auto stage = [&](uint32_t tile, uint4 (&regs)[(32 * DIM * 2 / 16 + 255) / 256], float &cn) {  // global -> registers
        constexpr uint32_t PIECES = 32 * DIM * 2 / 16;  // 16-byte pieces of the tile
#pragma unroll
        for (uint32_t i = 0; i < (PIECES + 255) / 256; ++i) {
            const uint32_t pc = t + 256 * i, row = pc / (DIM / 8), within = pc - row * (DIM / 8);
            const uint32_t j = 32 * tile + row;
            regs[i] = make_uint4(0u, 0u, 0u, 0u);
            if (pc < PIECES && j < k) regs[i] = *reinterpret_cast<const uint4 *>(cent_bf + (uint64_t)j * DIM + 8 * within);
        }
        cn = __builtin_inff();  // a row past the last list: never the minimum, never a candidate
        if (t < 32 && 32 * tile + t < k) cn = cnorm[32 * tile + t];
    };
    auto land = [&](uint32_t buf, const uint4 (&regs)[(32 * DIM * 2 / 16 + 255) / 256], float cn) {  // registers -> LDS
        constexpr uint32_t PIECES = 32 * DIM * 2 / 16;
        unsigned char *img = asg_lds + buf * (TILEB + 128);
#pragma unroll
        for (uint32_t i = 0; i < (PIECES + 255) / 256; ++i) {
            const uint32_t pc = t + 256 * i, row = pc / (DIM / 8), within = pc - row * (DIM / 8);
            if (pc < PIECES) *reinterpret_cast<uint4 *>(img + row * ROWB + 16 * within) = regs[i];
        }
        if (t < 32) reinterpret_cast<float *>(img + TILEB)[t] = cn;
    };
    float tmin[NT], thr[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) tmin[tl] = __builtin_inff(), thr[tl] = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {
        uint4 regs[(32 * DIM * 2 / 16 + 255) / 256];
        float cn_next;
        __syncthreads();  // the previous pass's last tile has been consumed
        stage(0, regs, cn_next);
        land(0, regs, cn_next);
        for (uint32_t tile = 0; tile < ntile; ++tile) {
            __syncthreads();  // tile `tile` is in LDS; the other buffer is free
            if (tile + 1 < ntile) stage(tile + 1, regs, cn_next);
            const unsigned char *img = asg_lds + (tile & 1u) * (TILEB + 128);
            const float *cnp = reinterpret_cast<const float *>(img + TILEB);
            asg_f32x16 acc[NT];
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) acc[tl] = asg_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const asg_bf16x8 af = *reinterpret_cast<const asg_bf16x8 *>(img + col * ROWB + (16 * m + 8 * kh) * 2);
#pragma unroll
                for (int tl = 0; tl < NT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfrag[tl][m], acc[tl], 0, 0, 0);
            }
            float cnr[16];  // |c|^2 of this lane's 16 rows: row = (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c4 = *reinterpret_cast<const float4 *>(cnp + 8 * g + 4 * kh);
                cnr[4 * g] = c4.x, cnr[4 * g + 1] = c4.y, cnr[4 * g + 2] = c4.z, cnr[4 * g + 3] = c4.w;
            }
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) {
                if (pass == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmin[tl] = fminf(tmin[tl], fmaf(-2.0f, acc[tl][r], cnr[r]));
                } else {
                    const uint64_t v = v0 + 32 * tl + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (fmaf(-2.0f, acc[tl][r], cnr[r]) <= thr[tl] && v < n) {
                            const uint32_t slot = atomicAdd(cand_cnt + v, 1u);
                            if (slot < RQ_ASSIGN_CAND) cand[v * RQ_ASSIGN_CAND + slot] = 32 * tile + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * kh;
                        }
                    }
                }
            }
            if (tile + 1 < ntile) land((tile + 1) & 1u, regs, cn_next);
        }
        if (pass == 0) {
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) {
                const float other = __shfl_xor(tmin[tl], 32, 64);  // the other 16 rows of every tile
                tmin[tl] = fminf(tmin[tl], other);
                const float rad = cmax + sqrtf(xn[tl]) * 1.000001f;
                const float mx = (0.00390625f + (float)(2 * DIM + 64) * 5.9604645e-8f) * 1.05f * (rad * rad);
                thr[tl] = tmin[tl] + 2.0f * mx;          // (inf / NaN -> no candidate -> the exact-order kernel takes the vector)
                thr[tl] = thr[tl] + fabsf(thr[tl]) * 1.0e-6f;  // the comparison's own rounding
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Coarse ranking through the same pre-filter (round 4; src/rabitq.rs:283-297: all k exact-order distances, select the nprobe
// smallest, sort them).  coarse_approx_kernel writes, for every query, a'_j = |c_j|^2 - 2 <c~_j, y~> (the approximation above minus
// the query's own |y|^2, a constant of the row) for every list; select_refine_wave_kernel (kernels_query.h) finds the nprobe-th
// smallest a' of a row, lists the lists within 2 m_y above it (the true nprobe nearest are among them: the nprobe smallest-a' lists
// all have e <= a' + |y|^2 + m, so the nprobe-th smallest exact distance is <= tau + m, and every list at or below it has
// a' + |y|^2 <= e + m <= tau + 2 m), recomputes THOSE in the reference's lane order and selects / sorts on the exact values.
// Roles are swapped against assign_approx_kernel (queries = A rows, lists = B columns), so that a wave's stores are 128-byte
// runs of one query's row.
// ------------------------------------------------------------------------------------------------
template <int W, int NT>
__global__ __launch_bounds__(256, (W <= 4 ? 2 : 1)) void coarse_approx_kernel(const float *__restrict__ y /* nq x dim, rotated queries */,
                                                              const uint16_t *__restrict__ cent_bf /* k x dim bf16 */,
                                                              const float *__restrict__ cnorm, uint32_t nq, uint32_t k,
                                                              float *__restrict__ dist /* nq x k */,
                                                              const uint16_t *__restrict__ y_bf /* nq x dim bf16 (to_bf16_kernel of y), or null */) {
    constexpr int DIM = 64 * W, NM = DIM / 16;
    constexpr uint32_t ROWB = DIM * 2 + 16, TILEB = 32 * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char asg_lds[];  // 2 x (tile image | 32 norms), as assign_approx_kernel
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 31, kh = lane >> 5;
    const uint32_t v0 = (blockIdx.x * 4 + wave) * (32 * NT);
    asg_bf16x8 afrag[NT][NM];  // k-elements 16 m + 8 kh .. + 7 of query v0 + 32 tile + col
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const uint32_t v = v0 + 32 * tl + col;
        const uint64_t vr = v < nq ? v : (nq - 1);
        if constexpr (NM > 32) {
            // dim > 512: the fragments come pre-rounded (to_bf16_kernel over the query rows, the rounding of asg_bf16_pair), one 16-byte
            // load per slab straight into its place.  Converted here, the f32 loads of all dim / 16 slabs were in flight beside the
            // fragments (384 + 192 registers at dim 768): 142 registers went to scratch memory in round 4 -- the only kernel of the
            // query path that needed any, and a dispatch that needs more scratch than the queue holds is set up and torn down around
            // the launch by the runtime (the 20 ms that appeared BETWEEN launches behind this kernel)
            const uint16_t *xb = y_bf + vr * DIM + 8 * kh;
#pragma unroll
            for (int m = 0; m < NM; ++m) afrag[tl][m] = __builtin_bit_cast(asg_bf16x8, *reinterpret_cast<const uint4 *>(xb + 16 * m));
        } else {
        const float *xp = y + vr * DIM + 8 * kh;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const float4 a = *reinterpret_cast<const float4 *>(xp + 16 * m), b = *reinterpret_cast<const float4 *>(xp + 16 * m + 4);
            const uint4 pk = make_uint4(asg_bf16_pair(a.x, a.y), asg_bf16_pair(a.z, a.w), asg_bf16_pair(b.x, b.y), asg_bf16_pair(b.z, b.w));
            afrag[tl][m] = __builtin_bit_cast(asg_bf16x8, pk);
        }
        }
    }
    const uint32_t ntile = (k + 31) / 32;
    constexpr uint32_t PIECES = 32 * DIM * 2 / 16, NREG = (PIECES + 255) / 256;
    auto stage = [&](uint32_t tile, uint4 (&regs)[NREG], float &cn) {  // global -> registers
#pragma unroll
        for (uint32_t i = 0; i < NREG; ++i) {
            const uint32_t pc = t + 256 * i, row = pc / (DIM / 8), within = pc - row * (DIM / 8);
            const uint32_t j = 32 * tile + row;
            regs[i] = make_uint4(0u, 0u, 0u, 0u);
            if (pc < PIECES && j < k) regs[i] = *reinterpret_cast<const uint4 *>(cent_bf + (uint64_t)j * DIM + 8 * within);
        }
        cn = 0.0f;
        if (t < 32 && 32 * tile + t < k) cn = cnorm[32 * tile + t];
    };
    auto land = [&](uint32_t buf, const uint4 (&regs)[NREG], float cn) {  // registers -> LDS
        unsigned char *img = asg_lds + buf * (TILEB + 128);
#pragma unroll
        for (uint32_t i = 0; i < NREG; ++i) {
            const uint32_t pc = t + 256 * i, row = pc / (DIM / 8), within = pc - row * (DIM / 8);
            if (pc < PIECES) *reinterpret_cast<uint4 *>(img + row * ROWB + 16 * within) = regs[i];
        }
        if (t < 32) reinterpret_cast<float *>(img + TILEB)[t] = cn;
    };
    uint4 regs[NREG];
    float cn_next;
    stage(0, regs, cn_next);
    land(0, regs, cn_next);
    for (uint32_t tile = 0; tile < ntile; ++tile) {
        __syncthreads();  // tile `tile` is in LDS; the other buffer is free
        if (tile + 1 < ntile) stage(tile + 1, regs, cn_next);
        const unsigned char *img = asg_lds + (tile & 1u) * (TILEB + 128);
        const float cn = reinterpret_cast<const float *>(img + TILEB)[col];  // |c|^2 of this lane's list
        asg_f32x16 acc[NT];
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = asg_f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const asg_bf16x8 cf = *reinterpret_cast<const asg_bf16x8 *>(img + col * ROWB + (16 * m + 8 * kh) * 2);
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[tl][m], cf, acc[tl], 0, 0, 0);
        }
        const uint32_t j = 32 * tile + col;
        if (j < k) {
#pragma unroll
            for (int tl = 0; tl < NT; ++tl)
#pragma unroll
                for (int r = 0; r < 16; ++r) {  // register r of lane half kh = query row (r & 3) + 8 (r >> 2) + 4 kh of the tile
                    const uint32_t v = v0 + 32 * tl + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4 * kh;
                    if (v < nq) dist[(uint64_t)v * k + j] = fmaf(-2.0f, acc[tl][r], cn);
                }
        }
        if (tile + 1 < ntile) land((tile + 1) & 1u, regs, cn_next);
    }
}

// exact-order distances (src/simd.rs:14-73) of every vector to its listed candidates, first minimum (smallest list id among
// equal distances: kmeans_nearest_cluster's strict `<` over ascending j); two lanes per vector (lane half hf = AVX lanes
// 4hf..4hf+3).  Vectors with 0 or more than RQ_ASSIGN_CAND candidates are appended to `redo` for the exact-order kernel.
__global__ __launch_bounds__(256) void assign_refine_kernel(const float *__restrict__ xrot, const float *__restrict__ centroids,
                                                            uint64_t n, uint32_t dim, const uint32_t *__restrict__ cand,
                                                            const uint32_t *__restrict__ cand_cnt, uint32_t *__restrict__ label,
                                                            float *__restrict__ dist, uint32_t *__restrict__ redo,
                                                            uint32_t *__restrict__ redo_cnt) {
    const uint32_t hf = threadIdx.x & 1;
    const uint64_t v = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 1;
    if (v >= n) return;
    const uint32_t nc = cand_cnt[v];
    if (nc == 0 || nc > RQ_ASSIGN_CAND) {
        if (hf == 0) redo[atomicAdd(redo_cnt, 1u)] = (uint32_t)v;
        return;
    }
    const float *x = xrot + v * dim + 4 * hf;
    float best = 0.0f;
    uint32_t lab = 0xFFFFFFFFu;
    for (uint32_t ci = 0; ci < nc; ++ci) {
        const uint32_t j = cand[v * RQ_ASSIGN_CAND + ci];
        const float *c = centroids + (uint64_t)j * dim + 4 * hf;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (uint32_t e = 0; e < dim; e += 8) {
            const float4 cv = *reinterpret_cast<const float4 *>(c + e), xv = *reinterpret_cast<const float4 *>(x + e);
            const float d0 = cv.x - xv.x, d1 = cv.y - xv.y, d2 = cv.z - xv.z, d3 = cv.w - xv.w;
            a0 = fmaf(d0, d0, a0), a1 = fmaf(d1, d1, a1), a2 = fmaf(d2, d2, a2), a3 = fmaf(d3, d3, a3);
        }
        const float c0 = a0 + __shfl_xor(a0, 1, 2), c1 = a1 + __shfl_xor(a1, 1, 2);
        const float c2 = a2 + __shfl_xor(a2, 1, 2), c3 = a3 + __shfl_xor(a3, 1, 2);
        const float dd = (c0 + c1) + (c2 + c3);
        if (lab == 0xFFFFFFFFu || dd < best || (dd == best && j < lab)) best = dd, lab = j;
    }
    if (hf == 0) {
        // the reference starts from (0, f32::MAX) and takes a list only if its distance is smaller: a minimum that is not
        // below f32::MAX (NaN cannot get here: such vectors have no candidate) keeps list 0
        label[v] = best < 3.402823466e+38f ? lab : 0u;
        dist[v] = best < 3.402823466e+38f ? best : 3.402823466e+38f;
    }
}
// results of the exact-order kernel for the listed vectors back to their rows
__global__ void assign_scatter_kernel(const uint32_t *__restrict__ redo, uint32_t m, const uint32_t *__restrict__ lab_in,
                                      const float *__restrict__ dist_in, uint32_t *__restrict__ label, float *__restrict__ dist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) label[redo[i]] = lab_in[i], dist[redo[i]] = dist_in[i];
}
