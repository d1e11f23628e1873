// common.h -- shared device helpers for the gfx950 RaBitQ kernels.
//
// Parity rules (SURVEY.md section 7 "hard parts" 3): the reference evaluates every f32 expression
// left to right with one rounding per operation (rustc never contracts), uses FMA only where
// src/simd.rs calls _mm256_fmadd_ps, and IEEE sqrt/div.  This translation unit is therefore built
// with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt, keeps f32 subnormals (hipcc
// default), and writes fmaf() explicitly where -- and only where -- the reference fuses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define RQ_WAVE 64

typedef float f32x2 __attribute__((ext_vector_type(2)));  // operands of v_pk_*_f32 (each component rounds on its own)

// ---- src/ord32.rs:12-26: monotone f32 <-> i32 key ----------------------------------------------
__host__ __device__ __forceinline__ int32_t ord32_from_f32(float x) {
    int32_t bits = __builtin_bit_cast(int32_t, x);
    uint32_t mask = ((uint32_t)(bits >> 31)) >> 1;
    return bits ^ (int32_t)mask;
}
__host__ __device__ __forceinline__ float ord32_to_f32(int32_t key) {
    uint32_t mask = ((uint32_t)(key >> 31)) >> 1;
    return __builtin_bit_cast(float, key ^ (int32_t)mask);
}
// unsigned-sortable form of the same key (for radix select / u64 composite keys)
__host__ __device__ __forceinline__ uint32_t ord32_biased(float x) {
    return (uint32_t)ord32_from_f32(x) ^ 0x80000000u;
}
__host__ __device__ __forceinline__ float ord32_unbias(uint32_t u) {
    return ord32_to_f32((int32_t)(u ^ 0x80000000u));
}

// ---- src/simd.rs:52-63: fold 8 AVX lanes held by 8 consecutive GPU lanes -------------------------
// ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)); float add is commutative, so an xor butterfly in the
// order 4, 1, 2 leaves the exact AVX result in all 8 lanes.
__device__ __forceinline__ float reduce8_lanes(float acc) {
    acc = acc + __shfl_xor(acc, 4, 8);
    acc = acc + __shfl_xor(acc, 1, 8);
    acc = acc + __shfl_xor(acc, 2, 8);
    return acc;
}
// same fold for 8 accumulators held in one thread
__device__ __forceinline__ float reduce8_regs(const float (&a)[8]) {
    float c0 = a[0] + a[4], c1 = a[1] + a[5], c2 = a[2] + a[6], c3 = a[3] + a[7];
    return (c0 + c1) + (c2 + c3);
}

// _mm256_cvtps_epi32 (src/simd.rs:215): round to nearest even; NaN / out of range -> 0x80000000
__device__ __forceinline__ int32_t cvtps_epi32(float x) {
    if (x >= -2147483648.0f && x < 2147483648.0f) return (int32_t)rintf(x);
    return (int32_t)0x80000000;
}

// Survivor record produced by the scan kernel and consumed by rerank / sort / replay.
struct __attribute__((aligned(16))) SurvRec {
    uint32_t pos;    // cluster-order position j (src/rabitq.rs:348)
    uint32_t slot;   // rank of the list in the query's probe order (src/rabitq.rs:304)
    float rough;     // src/rabitq.rs:352-363
    float accurate;  // filled by the rerank kernel (src/rerank.rs:85-90)
};
__device__ __forceinline__ uint64_t surv_key(const SurvRec &r) {
    return ((uint64_t)r.slot << 32) | r.pos;
}

// One wave-level append of the scan kernel: `cnt` (<= 64) survivors of 64 CONSECUTIVE list positions
// starting at `pos`, stored at recs[base .. base+cnt) in ascending position order (ballot rank order
// = lane order = position order).  Runs of different sub-tiles never interleave, so ordering a
// query's survivors in the reference's visiting order only needs the run directory sorted.
struct __attribute__((aligned(16))) RunRec {
    uint32_t pos;   // first list position of the 64-wide sub-tile
    uint32_t slot;  // probe slot
    uint32_t base;  // index of the run's first record in the query's survivor buffer
    uint32_t cnt;
};
__device__ __forceinline__ uint64_t surv_key(const RunRec &r) {
    return ((uint64_t)r.slot << 32) | r.pos;
}

// Where a query's survivor records and run descriptors live.  Uniform: query b owns slots [b * ucap, (b + 1) * ucap) of
// the pass's buffers.  Segmented (stages of a pass whose survivor counts are very unequal: the scan appends to a shared arena
// while counting per query, the exact counts size every query's segment, a scatter pass fills it): query b owns cap[b]
// slots from base[b].  The same geometry applies to the survivor
// records, the run directory and its scratch copy.
struct QSeg {
    const unsigned long long *base;  // per query: first slot; nullptr = uniform
    const uint32_t *cap;             // per query: slots
    uint32_t ucap;                   // uniform capacity (and the bound of the early stages of a segmented pass)
    __device__ __forceinline__ uint64_t at(uint32_t b) const { return base ? base[b] : (uint64_t)b * ucap; }
    __device__ __forceinline__ uint32_t capof(uint32_t b) const { return base ? cap[b] : ucap; }
};

// Per-(query, probe slot) scalars written by the prep kernel; 40 bytes = s_load_dwordx8 + x2.
struct __attribute__((aligned(8))) PairScalars {
    float lower;      // lower_bound                      (src/rabitq.rs:305)
    float delta;      // (hi - lo) * SCALAR               (:307)
    float sumq;       // scalar_sum as f32                (:322)
    float ycd;        // y_c_distance_square              (:319)
    float ycd_sqrt;   // dist_sqrt                        (:346)
    uint32_t row;         // query row b of this pair (pair id p = row * nprobe + slot)
    uint32_t list_begin;  // offsets[cluster]
    uint32_t list_len;
    uint32_t stream_begin;  // candidates the reference visits before this list (sum of earlier slots' lengths)
    uint32_t pad;
};

// Index-wide bounds of the Factor fields, used to decide when the integer-threshold form of the gate
// (matrix-core scan) is numerically safe for a query.
struct FactorStats {
    float cds_max, ppc_absmax, eb_max, invfip_absmax;
};

// The raw vectors the re-ranker gathers (src/rerank.rs:85-90; `base`, src/rabitq.rs:59, cluster order).  At
// 100M x 768 they are 307 GB, more than the HBM: part of them lives in pinned host memory that the kernels address
// directly over the host link (the reference's own answer to base > memory is a tiered store as well:
// crates/disk/src/cache.rs).  The split is PER LIST: the first h_c members of list c stay in HBM, its tail goes to
// the host tier.  Lists are ordered by centroid distance (src/rabitq.rs:232-238) and near-centroid vectors are the
// ones the re-ranker asks for most (a small center_distance_square lowers the distance to every query of the
// list), so the host tier holds the rows least likely to be gathered.  Small indexes have no host tier at all.
struct ListTier {  // one per list
    uint32_t off;        // first position of the list (offsets[c])
    uint32_t h;          // members kept in HBM
    uint32_t hbm_base;   // HBM rows of the lists before this one
    uint32_t host_base;  // host rows of the lists before this one
};
// Split rows (round 5): where no shadow rows fit beside the raw vectors (tiered indexes, and untiered ones that fill most of the
// HBM) every raw vector is stored as TWO 16-bit planes inside the row's own 4*dim bytes --
// first the dim upper halves of the f32 words rounded to nearest (h = (bits + 0x8000) >> 16: a bf16 image of the row with relative
// error <= 2^-8 per element), then the dim lower halves unchanged -- so that bits = (h << 16) + sext16(lo) restores every word
// exactly.  The re-ranker reads the first plane alone (half the bytes), proves most survivors out of the top-k with it
// (accurate_split_kernel, kernels_query.h) and fetches the second plane only for the rest.  Both tiers use the layout: over the
// host link the first plane alone is half the bytes as well.
struct RowRef {
    const float *p;  // the row's 4*dim bytes
    bool split;      // stored as two 16-bit planes
};
__host__ __device__ __forceinline__ uint32_t rq_split_join(uint32_t h16, uint32_t lo16) {
    return (h16 << 16) + (uint32_t)(int32_t)(int16_t)lo16;
}
// element e of a row / its store (cold paths: placement, dumps, shards, rq_rerank)
__host__ __device__ __forceinline__ float rq_row_get(const RowRef r, uint32_t dim, uint32_t e) {
    if (!r.split) return r.p[e];
    const uint16_t *h = reinterpret_cast<const uint16_t *>(r.p);
    return __builtin_bit_cast(float, rq_split_join(h[e], h[dim + e]));
}
__host__ __device__ __forceinline__ void rq_row_put(const RowRef r, uint32_t dim, uint32_t e, float v) {
    float *w = const_cast<float *>(r.p);
    if (!r.split) {
        w[e] = v;
        return;
    }
    const uint32_t bits = __builtin_bit_cast(uint32_t, v);
    uint16_t *h = reinterpret_cast<uint16_t *>(w);
    h[e] = (uint16_t)((bits + 0x8000u) >> 16), h[dim + e] = (uint16_t)bits;
}
struct BaseView {
    const float *dev;    // HBM tier
    const float *host;   // device-visible address of the pinned host tier; nullptr = every row is in HBM, at its position
    const ListTier *lt;  // k entries (tiered indexes only)
    uint32_t k;
    uint32_t split;      // the rows (of both tiers) are split rows
    // row of position p that belongs to list c
    __host__ __device__ __forceinline__ RowRef row_in_list(uint64_t p, const ListTier &t, uint32_t dim) const {
        const uint64_t local = p - t.off;
        if (local < t.h) return RowRef{dev + ((uint64_t)t.hbm_base + local) * dim, split != 0};
        return RowRef{host + ((uint64_t)t.host_base + (local - t.h)) * dim, split != 0};
    }
    // row of position p whose list is probe_row[slot] (the re-rankers: a survivor record names its slot)
    __device__ __forceinline__ RowRef row_of_slot(uint64_t p, const uint32_t *__restrict__ probe_row, uint32_t slot, uint32_t dim) const {
        if (!host) return RowRef{dev + p * dim, split != 0};
        return row_in_list(p, lt[probe_row[slot]], dim);
    }
    // row of position p, list unknown: bisection over the lists (cold paths: placement, dumps, shards, rq_rerank)
    __host__ __device__ __forceinline__ RowRef row(uint64_t p, uint32_t dim) const {
        if (!host) return RowRef{dev + p * dim, split != 0};
        uint32_t lo = 0, hi = k;  // largest c with lt[c].off <= p (empty lists share their start with the next one)
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (lt[mid].off <= p) lo = mid;
            else hi = mid;
        }
        return row_in_list(p, lt[lo], dim);
    }
};
