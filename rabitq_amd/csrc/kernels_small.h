// kernels_small.h -- RaBitQ::query (src/rabitq.rs:268-333) for SMALL batches (<= 64 queries, down to the one query per
// call of the reference's own harness, crates/cli/src/main.rs:69-80).
//
// A single query touches 50 MB (100M x 128, nprobe 64): ~8 us of HBM time.  The staged pipeline of kernels_query.h
// spends 0.25 ms on it, because its ~25 launches each are a chain of dependent memory round trips on a mostly idle chip.
// On this part a grid-wide barrier inside a launch (4-6 us) costs MORE than a kernel boundary (1.5-1.9 us), so the answer
// is not one cooperative kernel but FEW, FAT launches, each with as few dependent round trips as possible:
//
//   sb_front_kernel    rotate (src/utils.rs:237-258) + coarse distances (src/rabitq.rs:285-293), lists spread over blocks
//   sb_query_kernel    ONE 1024-thread block per query: probe selection (:294-297), per-list query quantisation (:304-317),
//                      stream offsets, and the early part of the candidate stream (the nearest list or so) scanned,
//                      re-ranked and replayed stage by stage INSIDE the block -- survivors, ranker state and thresholds
//                      never leave LDS, positions are visited in stream order so nothing has to be sorted -- then the work
//                      records of the final stage (or, for small indexes, the final stage itself and the results)
//   scan_kernel        the rest of the stream under the settled threshold: the existing whole-chip scan
//   sb_finish_kernel   re-rank + order + replay of the final stage's survivors, results, METRICS totals
//
// Every arithmetic routine is the one the large-batch path uses (select / prep / rough_distance / accurate_rows /
// replay_wave), so results are bit-identical to it and to the oracle; the staging (where a query's stream is cut) is
// different, which never changes a result.
#pragma once
#include "kernels_query.h"

#pragma clang fp contract(off)

#define RQ_SB_MAX_NQ 64u       // batches up to this size take the path
#define RQ_SB_QT 8             // queries per block of the front kernel
#define RQ_SB_LISTS 128u       // lists per block of the front kernel (two lanes per list)
#define RQ_SB_CAP 6144u        // survivor records a query's block keeps in LDS (96 KiB; one block per CU)
#define RQ_SB_TILE 1024u       // stream positions scanned per step (one per thread)
#define RQ_SB_MAX_STAGES 8
#define RQ_SB_MAX_TOPK 256u    // ranker state lives in LDS
#define RQ_SBF_RUNS 2048u      // sb_finish_kernel's LDS path: run descriptors ...
#define RQ_SBF_RECS 2048u      // ... and survivors of the final stage it holds
#define RQ_SB_MAX_K 8192u      // 16 wave slices of <= 512 lists each in the probe selection

// ------------------------------------------------------------------------------------------------
// sb_front_kernel: block (x = 128 lists, y = 8 queries).  Every block rotates its 8 queries itself (P is 4 dim^2 bytes,
// L2-resident; recomputing beats a launch boundary), then scores its 128 lists against them: two lanes per centroid row
// (lane half hf = AVX lanes 4hf..4hf+3, 16-byte loads), the row read once for all 8 queries.  Arithmetic: exactly
// rotate_valu_kernel's (src/simd.rs:257-314) and accurate_rows' (src/simd.rs:14-73; l2_squared_distance(centroid, y)).
// dynamic LDS: 2 * 8 * dim floats.  Block (0, 0) also zeroes the pass totals.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sb_front_kernel(const float *__restrict__ q, uint32_t len,
                                                       const float *__restrict__ P, const float *__restrict__ centroids,
                                                       float *__restrict__ y_out, float *__restrict__ qpad_out,
                                                       float *__restrict__ dist, uint32_t k, uint32_t dim, uint32_t nq,
                                                       unsigned long long *__restrict__ totals, uint32_t *__restrict__ big3) {
    extern __shared__ __attribute__((aligned(16))) float sbf[];
    float *xs = sbf, *ys = sbf + RQ_SB_QT * dim;
    const uint32_t q0 = blockIdx.y * RQ_SB_QT, t = threadIdx.x;
    const uint32_t nv = nq - q0 < RQ_SB_QT ? nq - q0 : RQ_SB_QT;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (t < 8) totals[t] = 0ull;
        if (t < 3) big3[t] = 0u;
    }
    for (uint32_t i = t; i < nv * dim; i += 256) {  // zero-pad to the padded dimension (src/rabitq.rs:277-280)
        const uint32_t v = i / dim, e = i - v * dim;
        xs[i] = e < len ? q[(uint64_t)(q0 + v) * len + e] : 0.0f;
    }
    __syncthreads();
    {  // rotation: lane <-> column j (coalesced reads of P), up to four queries per pass over the column
        const uint32_t ncg = dim >= 256 ? 1u : 256u / dim;  // thread groups working on different queries of the same columns
        for (uint32_t w = t; w < dim * ncg; w += 256) {
            const uint32_t g = w / dim, j = w - g * dim;
            for (uint32_t v0 = g; v0 < nv; v0 += 4 * ncg) {
                float acc[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int l = 0; l < 8; ++l) acc[u][l] = 0.0f;
                // 32 rows of the column in flight at a time (dim is a multiple of 64): the loop is bound by the round trips to
                // L2, not by its 4 x 32 fused multiply-adds per round
                for (uint32_t c0 = 0; c0 < dim; c0 += 32) {
                    float pv[32];
#pragma unroll
                    for (int l = 0; l < 32; ++l) pv[l] = P[(uint64_t)(c0 + l) * dim + j];
#pragma unroll
                    for (int cc = 0; cc < 32; cc += 8) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t v = v0 + u * ncg;
                            if (v < nv) {
#pragma unroll
                                for (int l = 0; l < 8; ++l) acc[u][l] = fmaf(xs[v * dim + c0 + cc + l], pv[cc + l], acc[u][l]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t v = v0 + u * ncg;
                    if (v < nv) ys[v * dim + j] = reduce8_regs(acc[u]);
                }
            }
        }
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (uint32_t i = t; i < nv * dim; i += 256) {
            y_out[(uint64_t)q0 * dim + i] = ys[i];
            qpad_out[(uint64_t)q0 * dim + i] = xs[i];
        }
    // coarse distances
    const uint32_t pair = t >> 1, hf = t & 1u, j = blockIdx.x * RQ_SB_LISTS + pair;
    const bool live = j < k;
    const float *row = centroids + (uint64_t)(live ? j : 0u) * dim + 4 * hf;
    float a[RQ_SB_QT][4];
#pragma unroll
    for (int v = 0; v < RQ_SB_QT; ++v) a[v][0] = a[v][1] = a[v][2] = a[v][3] = 0.0f;
    for (uint32_t c = 0; c < dim; c += 64) {
        float4 xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4 *>(row + c + 8 * u);
#pragma unroll
        for (int v = 0; v < RQ_SB_QT; ++v) {
            if ((uint32_t)v < nv) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 qv = *reinterpret_cast<const float4 *>(ys + v * dim + c + 8 * u + 4 * hf);
                    const float d0 = xv[u].x - qv.x, d1 = xv[u].y - qv.y, d2 = xv[u].z - qv.z, d3 = xv[u].w - qv.w;
                    a[v][0] = fmaf(d0, d0, a[v][0]), a[v][1] = fmaf(d1, d1, a[v][1]);
                    a[v][2] = fmaf(d2, d2, a[v][2]), a[v][3] = fmaf(d3, d3, a[v][3]);
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < RQ_SB_QT; ++v) {
        if ((uint32_t)v < nv) {
            const float c0 = a[v][0] + __shfl_xor(a[v][0], 1, 2), c1 = a[v][1] + __shfl_xor(a[v][1], 1, 2);
            const float c2 = a[v][2] + __shfl_xor(a[v][2], 1, 2), c3 = a[v][3] + __shfl_xor(a[v][3], 1, 2);
            if (live && hf == 0) dist[(uint64_t)(q0 + v) * k + j] = (c0 + c1) + (c2 + c3);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Probe selection, two levels (the same result as select_probe_wave: the nprobe smallest (Ord32 distance, list id)
// pairs, ascending).  Level 1: each of the 16 waves selects the `want` smallest of its slice of the distance row
// (<= 512 lists: 8 registers per lane) by bisection on the monotone u32 key, ties by list id, and sorts them across its
// lanes; level 2 (select_merge16): every candidate's global rank = its place in its own slice + the number of smaller
// keys in the other 15 sorted slices (binary searches in LDS) -- keys are unique, so ranks are a permutation and the
// winners land at their final positions.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void select_slice_wave(const float *__restrict__ d, uint32_t slen, uint32_t want, bool vec4,
                                                  uint32_t id_offset, unsigned long long *__restrict__ out /* 64, LDS */,
                                                  unsigned long long *win /* 64, LDS, this wave's */) {
    constexpr int KPL = 8;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t key[KPL];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    auto list_of = [&](int i) { return 256u * (uint32_t)(i >> 2) + 4u * lane + (uint32_t)(i & 3); };
#pragma unroll
    for (int i4 = 0; i4 < KPL; i4 += 4) {
        const uint32_t j0 = list_of(i4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec4 && j0 + 3 < slen) {
            v = *reinterpret_cast<const float4 *>(d + j0);
        } else {
            if (j0 < slen) v.x = d[j0];
            if (j0 + 1 < slen) v.y = d[j0 + 1];
            if (j0 + 2 < slen) v.z = d[j0 + 2];
            if (j0 + 3 < slen) v.w = d[j0 + 3];
        }
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i4 + e;
            key[i] = 0xFFFFFFFFu;  // "no list"
            if (j0 + e < slen) {
                key[i] = ord32_biased(ve[e]);
                kmin = key[i] < kmin ? key[i] : kmin;
                kmax = key[i] > kmax ? key[i] : kmax;
            }
        }
    }
    out[lane] = ~0ull;
    if (want == 0) return;  // wave-uniform
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
    // (the same in every lane after the butterfly: said explicitly so that the bisection below runs on scalar registers)
    kmin = __builtin_amdgcn_readfirstlane(kmin), kmax = __builtin_amdgcn_readfirstlane(kmax);
    uint32_t lo = kmin, hi = kmax, T = kmax;
    bool exact = false;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint32_t c = count_le(mid);
        if (c == want) {
            T = mid;
            exact = true;
            break;
        }
        if (c > want) hi = mid;
        else lo = mid + 1;
    }
    if (!exact) T = lo;
    uint32_t J = 0xFFFFFFFFu;  // among keys == T only (slice-local) ids <= J are taken
    if (!exact) {
        const uint32_t c_le = count_le(T);
        if (c_le > want) {  // ties at the threshold: the smallest list ids win
            const uint32_t c_lt = T ? count_le(T - 1) : 0u;
            const uint32_t need = want - c_lt;  // >= 1
            uint32_t jl = 0, jh = slen - 1;
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
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const uint32_t j = list_of(i);
        const bool take = j < slen && (key[i] < T || (key[i] == T && j <= J));
        const uint64_t m = __ballot(take);
        if (m) {
            if (take) win[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((unsigned long long)key[i] << 32) | (j + id_offset);
            base += (uint32_t)__popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    unsigned long long v = lane < want ? win[lane] : ~0ull;
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
    out[lane] = v;
}

// level 2, all 1024 threads (thread = candidate (w, i)); cand[16][64] sorted ascending per slice, padded with ~0
__device__ __forceinline__ void select_merge16(const unsigned long long (*cand)[64], uint32_t nprobe, uint32_t b,
                                               uint32_t *__restrict__ probe_cluster, float *__restrict__ probe_dist) {
    const uint32_t w = threadIdx.x >> 6, i = threadIdx.x & 63;
    const unsigned long long key = cand[w][i];
    if (key == ~0ull) return;
    uint32_t rank = i;
    uint32_t lo[16];  // per slice: the number of its keys below `key` (sixteen independent binary searches: their LDS reads overlap)
#pragma unroll
    for (int o = 0; o < 16; ++o) lo[o] = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1)
#pragma unroll
        for (int o = 0; o < 16; ++o)
            if (cand[o][lo[o] + step - 1] < key) lo[o] += step;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        if (lo[o] == 63 && cand[o][63] < key) lo[o] = 64;
        rank += (uint32_t)o == w ? 0u : lo[o];
    }
    if (rank >= nprobe) return;
    probe_cluster[(uint64_t)b * nprobe + rank] = (uint32_t)key;
    probe_dist[(uint64_t)b * nprobe + rank] = ord32_unbias((uint32_t)(key >> 32));
}

// accurate_rows for the block's LDS-resident survivors (same arithmetic: two lanes per row, lane half hf = AVX lanes
// 4hf..4hf+3 of src/simd.rs:14-73, chunks of 64 dimensions in order): up to dim 128 every load of a row is issued before
// the first one is used.
template <int W>
__device__ __forceinline__ void sb_accurate_rows(SurvRec *recs, uint32_t n, const BaseView &base, const float *q_lds,
                                                 uint32_t dim, const uint32_t *__restrict__ probe_row) {
    const uint32_t hf = threadIdx.x & 1;
    for (uint32_t i = threadIdx.x >> 1; i < n; i += 512) {
        const float *x;
        if (base.host == nullptr && !base.split) {
            x = base.dev + (uint64_t)recs[i].pos * dim + 4 * hf;
        } else {
            const RowRef rr = base.row_of_slot(recs[i].pos, probe_row, recs[i].slot, dim);
            if (rr.split) {
                const float r = exact_l2_pair_split(rr.p, q_lds, dim, hf);
                if (hf == 0) recs[i].accurate = r;
                continue;
            }
            x = rr.p + 4 * hf;
        }
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        auto chunk = [&](const float4 (&xv)[8], uint32_t c) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 qv = *reinterpret_cast<const float4 *>(q_lds + c + 8 * u + 4 * hf);
                const float d0 = xv[u].x - qv.x, d1 = xv[u].y - qv.y, d2 = xv[u].z - qv.z, d3 = xv[u].w - qv.w;
                a0 = fmaf(d0, d0, a0), a1 = fmaf(d1, d1, a1), a2 = fmaf(d2, d2, a2), a3 = fmaf(d3, d3, a3);
            }
        };
        if constexpr (W <= 2) {
            float4 xv[W][8];
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[w][u] = *reinterpret_cast<const float4 *>(x + 64 * w + 8 * u);
#pragma unroll
            for (int w = 0; w < W; ++w) chunk(xv[w], 64 * w);
        } else {
            for (uint32_t c = 0; c < dim; c += 64) {
                float4 xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const float4 *>(x + c + 8 * u);
                chunk(xv, c);
            }
        }
        const float c0 = a0 + __shfl_xor(a0, 1, 2), c1 = a1 + __shfl_xor(a1, 1, 2);
        const float c2 = a2 + __shfl_xor(a2, 1, 2), c3 = a3 + __shfl_xor(a3, 1, 2);
        const float r = (c0 + c1) + (c2 + c3);
        if (hf == 0) recs[i].accurate = r;
    }
}

// ------------------------------------------------------------------------------------------------
// sb_query_kernel
// ------------------------------------------------------------------------------------------------
struct SbArgs {
    // index
    const uint32_t *codes;
    const float4 *factors;
    const float *centroids;
    const uint32_t *offsets, *map_ids;
    BaseView base;
    // the pass's buffers
    const float *dist, *y, *qpad;  // coarse distances nq x k, rotated and padded queries nq x dim (sb_front_kernel)
    uint32_t *probe_cluster;
    float *probe_dist;
    PairScalars *scal;
    uint32_t *qnib;
    uint32_t *qf6;  // fp6 operand images, or null when no stage of the pass can use the matrix-core scan
    unsigned long long *rough_cnt, *surv_cnt, *totals;
    ReplayState rs;  // the pass's ranker state in global memory (handed to the final stage's kernels)
    float *out_dist;
    uint32_t *out_id, *out_n;
    uint32_t *recs;  // work records of the final stage (pair-major)
    FactorStats fs;
    uint32_t k, dim, nprobe, topk, cap, hcap;
    uint32_t nstages;  // stages scanned in the block
    uint32_t s_lo[RQ_SB_MAX_STAGES], s_hi[RQ_SB_MAX_STAGES];
    uint32_t finalize;    // 1: the stream ends in the block: results and totals are written here (heap ranker)
    uint32_t fill_final;  // 1: the pair-major work records of the stage [final_lo, end) are written here
    uint32_t final_lo;
    unsigned long long *stamps;  // developer hook (scan_debug bit 4096): block 0 records the 100 MHz clock at its phase boundaries
};

// results of a finished query (src/rerank.rs:108-113: the heap's Vec order) and its share of the pass totals
// (metrics_sum_kernel's sums), by ONE wave; the state is read through `st` at row sb (global, or this block's LDS image)
__device__ __forceinline__ void sb_write_results(const ReplayState &st, uint32_t sb, uint32_t b, uint32_t topk,
                                                 const uint32_t *__restrict__ map_ids, float *__restrict__ out_dist,
                                                 uint32_t *__restrict__ out_id, uint32_t *__restrict__ out_n,
                                                 unsigned long long rough, uint32_t cap, unsigned long long *__restrict__ totals) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t len = st.heap_len[sb];
    for (uint32_t e = lane; e < len; e += 64) {
        out_dist[(uint64_t)b * topk + e] = ord32_to_f32(st.heap_key[(uint64_t)sb * topk + e]);
        out_id[(uint64_t)b * topk + e] = map_ids[st.heap_id[(uint64_t)sb * topk + e]];  // position -> original id
    }
    if (lane == 0) {
        out_n[b] = len;
        const uint32_t need = st.need[sb];
        const bool ok = !st.ovf[sb];
        atomicAdd(totals + 0, rough);
        if (ok) atomicAdd(totals + 1, (unsigned long long)st.precise[sb]);
        else atomicAdd(totals + 2, 1ull);
        atomicAdd(totals + 3, (unsigned long long)st.nsurv[sb]);
        atomicMax(totals + 4, (unsigned long long)need);
    }
}

// MODE 0: heap ranker, heap in LDS; 1: heap ranker, heap in a register pair (topk < 64); 2: heuristic ranker
template <int W, int MODE>
__global__ __launch_bounds__(1024) void sb_query_kernel(const SbArgs a) {
    constexpr bool HEUR = MODE == 2, REGHEAP = MODE == 1;
    constexpr int LP = W == 1 ? 16 : (W == 2 ? 32 : 64), R = W <= 4 ? 1 : W / 4;
    static_assert(4 * LP * R == 64 * W, "prep_small_pairs geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char sbq_raw[];  // recs[RQ_SB_CAP] | qv[dim] | heap state 2 x topk | heap work 2 x topk
    __shared__ unsigned long long cand[16][64];
    __shared__ unsigned long long win[16][64];
    __shared__ __attribute__((aligned(16))) uint32_t s_qn[8 * W];
    __shared__ __attribute__((aligned(16))) uint32_t s_pl[4][2 * W];  // the query's four bit planes (src/simd.rs:83-107), dword w <-> dimensions 32w..32w+31
    __shared__ uint32_t wcnt[4][16];
    __shared__ float s_thr, s_recent;
    __shared__ uint32_t s_hlen, s_precise, s_need, s_nsurv, s_nshadow, s_wcount, s_alen, s_ovf;
    const uint32_t b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t k = a.k, dim = a.dim, nprobe = a.nprobe, topk = a.topk;
    uint32_t n_stamp = 0;
    auto stamp = [&]() {
        if (a.stamps && b == 0 && t == 0 && n_stamp < 30) a.stamps[1 + n_stamp] = wall_clock64(), a.stamps[0] = ++n_stamp;
    };
    stamp();
    SurvRec *recs = reinterpret_cast<SurvRec *>(sbq_raw);
    float *qv = reinterpret_cast<float *>(recs + RQ_SB_CAP);
    int32_t *hk_state = reinterpret_cast<int32_t *>(qv + dim);
    uint32_t *hi_state = reinterpret_cast<uint32_t *>(hk_state + topk);
    int32_t *hk_work = reinterpret_cast<int32_t *>(hi_state + topk);
    uint32_t *hi_work = reinterpret_cast<uint32_t *>(hk_work + topk);

    // ---- probe selection (src/rabitq.rs:294-297) ------------------------------------------------------------------
    {
        const uint32_t S = (((k + 15u) / 16u) + 3u) & ~3u;  // lists per wave slice (<= 512 for k <= 8192), a multiple of 4
        const uint32_t lo = wave * S;
        const uint32_t slen = lo < k ? (k - lo < S ? k - lo : S) : 0u;
        select_slice_wave(a.dist + (uint64_t)b * k + lo, slen, nprobe < slen ? nprobe : slen, (k & 3u) == 0u, lo, cand[wave], win[wave]);
    }
    __syncthreads();
    stamp();
    select_merge16(cand, nprobe, b, a.probe_cluster, a.probe_dist);
    for (uint32_t c = t * 4; c < dim; c += 4096) *reinterpret_cast<float4 *>(qv + c) = *reinterpret_cast<const float4 *>(a.qpad + (uint64_t)b * dim + c);
    if (t == 0) {  // ranker state of a fresh query (src/rerank.rs:70-77, :129-139)
        s_thr = 3.402823466e+38f, s_recent = -3.402823466e+38f;
        s_hlen = 0, s_precise = 0, s_need = 0, s_nsurv = 0, s_nshadow = 0, s_wcount = 0, s_alen = 0, s_ovf = 0;
        a.surv_cnt[b] = 0ull;
    }
    __syncthreads();
    stamp();
    // ---- per-list query quantisation (:304-317), stream offsets ------------------------------------------------------
    {
        constexpr uint32_t PPW = 64 / LP;
        for (uint32_t g = wave; g * PPW < nprobe; g += 16)
            prep_small_pairs<LP, R, 1>(a.y, a.centroids, a.offsets, a.probe_cluster, a.probe_dist, (b + 1) * nprobe, nprobe, a.scal,
                                       a.qnib, a.qf6, k, 1u, b * nprobe + g * PPW + lane / LP);
    }
    __syncthreads();
    stamp();
    if (wave == 0) pair_prefix_row(a.scal, b, nprobe, a.rough_cnt);
    __syncthreads();
    stamp();

    // ---- the early part of the stream, stage by stage, all in LDS -----------------------------------------------------
    ReplayState ls;
    ls.thr = &s_thr, ls.heap_len = &s_hlen, ls.heap_key = hk_state, ls.heap_id = hi_state, ls.precise = &s_precise;
    ls.need = &s_need, ls.ovf = &s_ovf, ls.nsurv = &s_nsurv, ls.nshadow = &s_nshadow, ls.recent_max = &s_recent, ls.win_count = &s_wcount;
    ls.arr_len = &s_alen, ls.arr = a.rs.arr + (uint64_t)b * a.hcap, ls.hcap = a.hcap;
    uint32_t n = 0;  // survivors waiting in `recs` (block-uniform)
    // re-rank the waiting survivors (src/rerank.rs:85-90) and replay the ranker over them (:81-106 / :143-168); they
    // are in visiting order already.  Cutting the stream here is one more stage boundary: never changes a result.
    auto flush = [&]() {
        if (n == 0) return;
        __syncthreads();  // every record written so far is in LDS
        stamp();
        sb_accurate_rows<W>(recs, n, a.base, qv, dim, a.probe_cluster + (uint64_t)b * nprobe);
        __syncthreads();
        stamp();
        if (wave == 0) {
            replay_wave<HEUR, REGHEAP, true>(recs, nullptr, n, topk, 0u, ls, hk_work, hi_work);
            if (lane == 0) s_nsurv += n;  // (s_need stays 0: these survivors never touch the pass's global buffers)
        }
        __syncthreads();
        stamp();
        n = 0;
    };
    // candidates per thread and step: their code / factor loads are all in flight before the first one is scored (a
    // lone block on an idle chip pays a full memory round trip per dependent step, so steps must be few and fat)
    constexpr int CPT = W <= 2 ? 4 : (W <= 4 ? 2 : 1);
    constexpr uint32_t STEP = CPT * RQ_SB_TILE;
    for (uint32_t sg = 0; sg < a.nstages; ++sg) {
        const uint32_t s_lo = a.s_lo[sg], s_hi = a.s_hi[sg];
        for (uint32_t slot = 0; slot < nprobe; ++slot) {
            PairScalars ps = a.scal[(uint64_t)b * nprobe + slot];
            {  // the same record in every lane: held in scalar registers, so that the loops below are scalar-controlled and the
               // query-side terms of the rough distance enter the vector ops as scalar operands
                auto uf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
                ps.lower = uf(ps.lower), ps.delta = uf(ps.delta), ps.sumq = uf(ps.sumq), ps.ycd = uf(ps.ycd), ps.ycd_sqrt = uf(ps.ycd_sqrt);
                ps.list_begin = __builtin_amdgcn_readfirstlane(ps.list_begin), ps.list_len = __builtin_amdgcn_readfirstlane(ps.list_len);
                ps.stream_begin = __builtin_amdgcn_readfirstlane(ps.stream_begin);
            }
            if (ps.list_len == 0) continue;
            if (ps.stream_begin >= s_hi) break;
            if ((uint64_t)ps.stream_begin + ps.list_len <= s_lo) continue;
            const uint32_t lo = s_lo > ps.stream_begin ? s_lo - ps.stream_begin : 0u;
            uint32_t hi = s_hi - ps.stream_begin;
            hi = hi < ps.list_len ? hi : ps.list_len;
            if (t < 8 * W) s_qn[t] = a.qnib[((uint64_t)b * nprobe + slot) * (8 * W) + t];
            __syncthreads();
            // One CU scans alone here, so the scoring is kept as cheap as the data allows: the asymmetric dot product in its
            // AND + popcount form (src/utils.rs:113-135: sum_p popcount(code & plane_p) << p), 2 VALU ops per (plane, code
            // dword) instead of the 7 of nibble expansion + v_dot8.  The bit planes come from the 4-bit codes once per list.
            if (t < 8 * W) {
                const uint32_t p = t / (2 * W), w = t - p * (2 * W);
                uint32_t word = 0;
#pragma unroll
                for (int j = 0; j < 32; ++j) word |= ((s_qn[4 * w + (j >> 3)] >> (4 * (j & 7) + p)) & 1u) << j;
                s_pl[p][w] = word;
            }
            __syncthreads();
            constexpr bool PL_REGS = W <= 2;  // planes in registers for narrow vectors, broadcast LDS reads beyond
            uint32_t pl[PL_REGS ? 4 : 1][PL_REGS ? 2 * W : 1];
            if constexpr (PL_REGS) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int w = 0; w < 2 * W; ++w) pl[p][w] = s_pl[p][w];
            }
            auto plane = [&](int p, int w) -> uint32_t {
                if constexpr (PL_REGS) return pl[p][w];
                else return s_pl[p][w];
            };
            for (uint32_t p0 = lo; p0 < hi; p0 += STEP) {
                if (n + STEP > RQ_SB_CAP) flush();  // room for whatever this step lets through (block-uniform)
                // the threshold the ranker holds now: an upper bound of the reference's for what follows
                const float thr = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_thr)));
                uint32_t pos[CPT], code[CPT][2 * W];
                float4 fac[CPT];
                bool in[CPT];
#pragma unroll
                for (int c = 0; c < CPT; ++c) {  // sub-tile c: positions p0 + 1024 c + t
                    const uint32_t p = p0 + c * RQ_SB_TILE + t;
                    in[c] = p < hi;
                    pos[c] = ps.list_begin + (in[c] ? p : lo);
                    const uint32_t *cp = a.codes + (uint64_t)pos[c] * (2 * W);
                    if constexpr ((2 * W) % 4 == 0) {
#pragma unroll
                        for (int i = 0; i < 2 * W; i += 4) {
                            const uint4 v = *reinterpret_cast<const uint4 *>(cp + i);
                            code[c][i] = v.x, code[c][i + 1] = v.y, code[c][i + 2] = v.z, code[c][i + 3] = v.w;
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 2 * W; i += 2) {
                            const uint2 v = *reinterpret_cast<const uint2 *>(cp + i);
                            code[c][i] = v.x, code[c][i + 1] = v.y;
                        }
                    }
                    fac[c] = a.factors[pos[c]];
                }
                float rough[CPT];
                uint64_t m[CPT];
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    uint32_t cp[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (int i = 0; i < 2 * W; ++i)
#pragma unroll
                        for (int p = 0; p < 4; ++p) cp[p] += (uint32_t)__builtin_popcount(code[c][i] & plane(p, i));
                    const uint32_t acc = cp[0] + (cp[1] << 1) + (cp[2] << 2) + (cp[3] << 3);
                    rough[c] = rough_distance(acc, fac[c], ps.lower, ps.delta, ps.sumq, ps.ycd, ps.ycd_sqrt);  // src/rabitq.rs:352-363
                    m[c] = __ballot(in[c] && rough[c] < thr);                                                   // src/rerank.rs:84
                    if (lane == 0) wcnt[c][wave] = (uint32_t)__popcll(m[c]);
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < CPT; ++c) {  // survivors in position order: sub-tile, wave, lane
                    uint32_t woff = 0, total = 0;
#pragma unroll
                    for (int w2 = 0; w2 < 16; ++w2) {
                        const uint32_t cn = wcnt[c][w2];
                        woff += (uint32_t)w2 < wave ? cn : 0u;
                        total += cn;
                    }
                    if ((m[c] >> lane) & 1ull) {
                        SurvRec r;
                        r.pos = pos[c], r.slot = slot, r.rough = rough[c], r.accurate = 0.0f;
                        recs[n + woff + (uint32_t)__popcll(m[c] & ((1ull << lane) - 1ull))] = r;
                    }
                    n += __builtin_amdgcn_readfirstlane(total);
                }
                __syncthreads();
            }
        }
        flush();  // the stage ends: the next one starts under the threshold learnt here
    }

    // ---- hand-over ------------------------------------------------------------------------------------------------------
    stamp();
    if (a.finalize && !HEUR) {
        if (wave == 0) sb_write_results(ls, 0u, b, topk, a.map_ids, a.out_dist, a.out_id, a.out_n, a.rough_cnt[b], a.cap, a.totals);
        return;
    }
    for (uint32_t e = t; e < s_hlen; e += 1024) {
        a.rs.heap_key[(uint64_t)b * topk + e] = hk_state[e];
        a.rs.heap_id[(uint64_t)b * topk + e] = hi_state[e];
    }
    if (t == 0) {
        a.rs.thr[b] = s_thr, a.rs.recent_max[b] = s_recent;
        a.rs.heap_len[b] = s_hlen, a.rs.precise[b] = s_precise, a.rs.need[b] = s_need, a.rs.nsurv[b] = s_nsurv;
        a.rs.nshadow[b] = 0, a.rs.win_count[b] = s_wcount, a.rs.arr_len[b] = s_alen, a.rs.ovf[b] = 0;
    }
    if (a.fill_final) {  // the final stage's pair-major work records: what stage_fill_kernel would write
        __syncthreads();
        for (uint32_t slot = t >> 4; slot < nprobe; slot += 64)
            stage_fill_item(a.scal, a.probe_cluster, a.qnib, a.rs.thr, b * nprobe + slot, nprobe, nprobe, 8 * W, a.final_lo, 0xFFFFFFFFu,
                            0u, nullptr, nullptr, a.recs, a.fs, 0u, nullptr, nullptr, k, nullptr);
    }
}

// ------------------------------------------------------------------------------------------------
// sb_finish_kernel: the final stage's survivors of one query (block): exact distances, run directory ordered, replay
// (stage_finish_kernel<false>), then the results and the query's share of the totals (finalize_heap_kernel +
// metrics_sum_kernel).  Heap ranker only.
// ------------------------------------------------------------------------------------------------
template <bool REGHEAP>
__global__ __launch_bounds__(1024) void sb_finish_kernel(SurvRec *__restrict__ surv, RunRec *__restrict__ runs,
                                                         unsigned long long *__restrict__ surv_cnt, const QSeg seg,
                                                         const BaseView base, const float *__restrict__ qpad, uint32_t dim,
                                                         uint32_t topk, ReplayState st, const uint32_t *__restrict__ probe_cluster,
                                                         uint32_t nprobe, uint32_t presorted, const uint32_t *__restrict__ map_ids,
                                                         float *__restrict__ out_dist, uint32_t *__restrict__ out_id,
                                                         uint32_t *__restrict__ out_n, const unsigned long long *__restrict__ rough_cnt,
                                                         unsigned long long *__restrict__ totals) {
    __shared__ int32_t hkey[REGHEAP ? 1 : RQ_MAX_TOPK];
    __shared__ uint32_t hid[REGHEAP ? 1 : RQ_MAX_TOPK];
    extern __shared__ __attribute__((aligned(16))) float fin_q[];  // dim floats: the padded query | 2 x RQ_SBF_RUNS descriptors | RQ_SBF_RECS records
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
        surv_cnt[b] = 0;
    }
    if (n) {  // block-uniform
        SurvRec *recs = surv + qat;
        for (uint32_t c = threadIdx.x * 4; c < dim; c += blockDim.x * 4)
            *reinterpret_cast<float4 *>(fin_q + c) = *reinterpret_cast<const float4 *>(qpad + (uint64_t)b * dim + c);
        __syncthreads();
        if (!(presorted & 2u))  // bit 1: the exact distances were computed by a whole-chip launch already
            accurate_rows(recs, n, base, fin_q, dim, threadIdx.x >> 1, blockDim.x >> 1, probe_cluster + (uint64_t)b * nprobe);
        if (nruns <= RQ_SBF_RUNS && n <= RQ_SBF_RECS && nprobe <= 64) {
            // The usual case, all in LDS: the run directory is ordered by (probe slot, position) with a counting sort on the
            // slot and rank counting inside a slot's bucket (four barriers instead of a bitonic network's fifty), the
            // survivors are gathered into visiting order by the whole block, and the ranker replays over contiguous LDS
            // records -- no global round trip inside the serial replay loop.
            RunRec *R = reinterpret_cast<RunRec *>(fin_q + dim), *B = R + RQ_SBF_RUNS;
            SurvRec *L = reinterpret_cast<SurvRec *>(B + RQ_SBF_RUNS);
            uint32_t *off = reinterpret_cast<uint32_t *>(B);  // (after the sort: exclusive prefix sums of the runs' counts)
            __shared__ uint32_t hist[64], bstart[65], bcur[64], wsum[16];
            const uint32_t t = threadIdx.x, nt = blockDim.x;
            __syncthreads();  // (accurate_rows' stores are visible to the gather below)
            if (t < 64) hist[t] = 0;
            __syncthreads();
            for (uint32_t i = t; i < nruns; i += nt) {
                R[i] = runs[qat + i];
                atomicAdd(&hist[R[i].slot & 63u], 1u);
            }
            __syncthreads();
            if (t < 64) {
                const uint32_t h = hist[t];
                uint32_t incl = h;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o, 64);
                    if ((int)t >= o) incl += up;
                }
                bstart[t] = incl - h, bcur[t] = incl - h;
                if (t == 63) bstart[64] = incl;
            }
            __syncthreads();
            for (uint32_t i = t; i < nruns; i += nt) B[atomicAdd(&bcur[R[i].slot & 63u], 1u)] = R[i];
            __syncthreads();
            for (uint32_t j = t; j < nruns; j += nt) {  // rank inside the slot's bucket: positions are unique there
                const RunRec me = B[j];
                const uint32_t lo = bstart[me.slot & 63u], hi = bstart[(me.slot & 63u) + 1];
                uint32_t rank = 0;
                for (uint32_t q2 = lo; q2 < hi; ++q2) rank += B[q2].pos < me.pos ? 1u : 0u;
                R[lo + rank] = me;
            }
            __syncthreads();
            {  // exclusive prefix sums of the ordered runs' counts (B is free now): every thread owns a contiguous stretch
                const uint32_t per = (nruns + nt - 1) / nt, i0 = t * per < nruns ? t * per : nruns, i1 = i0 + per < nruns ? i0 + per : nruns;
                uint32_t sum = 0;
                for (uint32_t i = i0; i < i1; ++i) sum += R[i].cnt;
                uint32_t incl = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o, 64);
                    if ((int)(t & 63) >= o) incl += up;
                }
                if ((t & 63) == 63) wsum[t >> 6] = incl;
                __syncthreads();
                uint32_t run = incl - sum;
                for (uint32_t w = 0; w < (t >> 6); ++w) run += wsum[w];
                for (uint32_t i = i0; i < i1; ++i) {
                    off[i] = run;
                    run += R[i].cnt;
                }
            }
            __syncthreads();
            for (uint32_t e = t; e < n; e += nt) {  // record e of the visiting order: its run by bisection over the prefix sums
                uint32_t lo = 0, hi = nruns;  // largest r with off[r] <= e
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (off[mid] <= e) lo = mid;
                    else hi = mid;
                }
                L[e] = recs[R[lo].base + (e - off[lo])];
            }
            __syncthreads();
            if (t < 64) replay_wave<false, REGHEAP, true>(L, nullptr, n, topk, b, st, hkey, hid);
        } else {
            if (nruns <= RQ_SORT_LDS_RECS || !(presorted & 1u)) sort_segment(runs + qat, nruns);
            __syncthreads();
            if (threadIdx.x < 64) replay_wave<false, REGHEAP>(recs, runs + qat, nruns, topk, b, st, hkey, hid);
        }
    }
    __threadfence_block();  // the state lane 0 (and the heap's lanes) stored is read back by the other lanes of the wave below
    if (threadIdx.x < 64)  // the same wave that wrote the state (and thread 0's counter updates above): program order
        sb_write_results(st, b, b, topk, map_ids, out_dist, out_id, out_n, rough_cnt[b], cap, totals);
}
