// kernels_scan_common.h -- what the scan kernels share with the kernels that prepare their work: the per-stage record layout,
// the launch argument blocks, the survivor-arena reservation, the reference's rough-distance expression.  No kernel is defined
// here, so the translation units of the scan kernels (inst_scan_*.hip) can include it without dragging the others in.
#pragma once
#include "common.h"

// Per-stage work records.  Everything the scan needs about one (query, list) pair, contiguous, so
// that the scan's inner loop is one pointer bump plus immediate-offset scalar loads:
//   dwords [0, opdw)    query operand: 4-bit codes 8 per dword (fused kernel, 8W), the 4 bit planes (8W), or
//                       the fp6 image of the codes (matrix-core kernel, 12W)
//   dwords opdw + ...   RQ_REC_* below
// Cluster-major: records of the pairs probing list c are stored at grp_start[c] ...; pair-major:
// record i belongs to pair i (pairs outside the stage get an empty range).
#define RQ_REC_LOWER 0
#define RQ_REC_DELTA 1
#define RQ_REC_SUMQ 2
#define RQ_REC_YCD 3
#define RQ_REC_YCD_SQRT 4
#define RQ_REC_THR 5
#define RQ_REC_LO 6          // first list position of this pair that belongs to the stage
#define RQ_REC_HI 7          // one past the last
#define RQ_REC_ROW 8
#define RQ_REC_SLOT 9
#define RQ_REC_LIST_BEGIN 10
#define RQ_REC_LIST_LEN 11
#define RQ_REC_V0 12         // 8 dwords: per-query bf16 operand of the integer threshold S*(c,q) = sum_r u'_c[r] * v'_q[r]
#define RQ_REC_CELL0 12      // record-major (VALU) records only: directory cell of this pair's list position 0 (dense directories)
#define RQ_REC_TAIL 20
// Tile images of the ADDITIVE-gate matrix-core scan (scan_mfma_kernel<.., ADD = true>): the tail stops before the bf16
// threshold operand (12 dwords per row), and the 32 rows' accumulator start values C_q follow the tails as ONE 32-float
// array (the kernel reads them as the C operand of its first MFMA: four broadcast 16-byte LDS reads per tile)
#define RQ_RECA_TAIL 12
// row stride (dwords) of a tile image's operand rows: opdw + 2 keeps 8-byte reads of the rows conflict-free; the additive format
// pads to opdw + 4, so that rows of 24 dwords (dim 128) are 16-byte aligned (28 = 4 x 7: three conflict-free ds_read_b128 per lane)
__host__ __device__ constexpr uint32_t rq_img_opld(uint32_t opdw, bool additive) { return opdw + (additive ? 4u : 2u); }
__host__ __device__ constexpr uint32_t rq_img_dwords(uint32_t opdw, bool additive) {
    return 32u * rq_img_opld(opdw, additive) + (additive ? 32u * RQ_RECA_TAIL + 32u : 32u * RQ_REC_TAIL);
}


// ------------------------------------------------------------------------------------------------
// THE SCAN: calculate_rough_distance (src/rabitq.rs:336-367) + asymmetric_binary_dot_product
// (src/utils.rs:113-135) + binary_dot_product (src/simd.rs:326-384), fused with the re-rank gate
// `rough < threshold` (src/rerank.rs:84).
//
//   s      = sum_p popcount(code & plane_p) << p                      (u32, exact)
//   rough  = ((cds + ycd) + lo*ppc) + (((2*s - sumq) * fip) * delta) - eb * sqrt(ycd)
//            evaluated left to right, one rounding per op, no contraction.
//
// One 256-thread block = one tile of 256*CPL consecutive list positions of one group.  Each lane
// keeps CPL candidates (W u64 code words + the 16-byte Factor) in registers; the block then loops
// over the group's (query, slot) pairs, whose operands are wave-uniform (SGPR / scalar loads).
// HBM traffic is the list itself: D/8 + 16 bytes per candidate, 16 B/lane coalesced loads at D=128.
// ------------------------------------------------------------------------------------------------
// Timing ablations and cycle counters of the scan kernels change results (or cost registers in the hot loops): they are compiled
// only into the developer build (-DRQ_DEV_ABLATIONS -> librabitq_hip_dev.so, used by scripts/exp/*); the shipped library's
// kernels carry none of these branches and rq_set_option rejects the bits.
#ifdef RQ_DEV_ABLATIONS
#define RQ_DBG(args, bits) ((args).dbg & (bits))
#else
#define RQ_DBG(args, bits) 0u
#endif
struct ScanArgs {   // scalars only; pointers are explicit __restrict__ kernel parameters so the
                    // compiler keeps the wave-uniform operand fetches on the scalar unit (s_load)
    uint32_t cap, tiles_per_group, ngroups, cluster_major;
    uint32_t dbg;  // developer ablations (scripts/ablate_scan.py); 0 in production
    // A stage whose grid would exceed the launch bound is issued as several launches: this one covers groups
    // group_base .. and, per group, tiles tile_base .. tile_base + tiles_per_group (host: launch_scan_chunks)
    uint32_t group_base, tile_base;
    // use_table: block b scans entry group_base + b of the index's tile table (one entry per existing (list, tile):
    // unbalanced lists launch no empty blocks, and the list's bounds arrive with the entry instead of a second
    // dependent load); ngroups then counts table entries and tiles_per_group is 1
    uint32_t use_table;
    // dense_dir: the stage's run descriptors go to a directory indexed by stream position (cell = the record's
    // RQ_REC_CELL0 + list position / 64: already in the reference's visiting order, nothing to sort) instead of being
    // appended in completion order
    uint32_t dense_dir;
    // per-query segments / arena mode: everything about them lives in a ScanExtra in device memory (nullptr: the uniform
    // geometry, records straight to the query's buffer) and is read on the survivor path only -- the hot loops keep their
    // kernel arguments in scalar registers, and a fat argument block costs them spills
    const struct ScanExtra *x;
};
// Survivor geometry beyond the uniform one.
// Arena mode (stages of a large batch that can exceed the uniform capacity): the survivors of a stage are first appended,
// in no particular order, to ONE arena shared by all queries (RQ_ARENA_SHARDS shards, each with its own 64-bit cursor
// -- records | runs << 32 --, chosen by block id: an append costs one more, uncontended atomic), while surv_cnt only
// COUNTS per query -- and what that count returns is the run's place inside its query's future segment (arena_places); the
// exact counts then size a segment per query and arena_scatter_kernel moves every run to its place.  Nothing is sized for a
// worst query, and the scatter needs no atomics (one per run -- 2.5e8 per step on the hard benchmark distribution -- was
// most of its time; issued from the scan's flushes they ride on a round trip the flush waits for anyway).
struct ScanExtra {
    uint2 *arena_places;        // per run descriptor: {first record, directory slot} of the run INSIDE its query's segment -- the value the
                                // scan's per-query count returned, so that the scatter pass needs no reservation of its own
    const uint32_t *reserved;
    SurvRec *arena_recs;        // nullptr: records go straight to the query's segment
    uint4 *arena_runs;          // {pos, slot | cnt << 16, query, record offset in the arena}
    unsigned long long *arena_cur;  // RQ_ARENA_SHARDS cursors, [SHARDS] overflow flag, [SHARDS + 1] (host), [SHARDS + 2] cursor of the common area
    unsigned int *arena_fail;   // per shard: run index of the first append it turned away (0xFFFFFFFF: none)
    uint32_t arena_sub, arena_rsub;  // capacity of a shard: records, runs (the same for both arrays)
    uint32_t arena_common;      // capacity of the common area behind the shards (records = runs), for what a full shard turns away
    uint32_t pad;
};
#define RQ_ARENA_SHARDS 2048u
#define RQ_ARENA_COMMON_BLOCKS 512u
// Reserve `nrec` records + `nrun` run descriptors of the arena for the calling lane's block: in the block's shard, or --
// when that is full (few, heavy blocks) -- in the common area.  Returns false if neither has room (the overflow flag is
// set: the host doubles the arena and repeats the stage).  rec_off / run_off: indices into arena_recs / arena_runs.
// The 64 bytes of a ScanExtra through ONE scalar load (the pointer is a kernel argument: wave-uniform).  Left to the compiler the
// struct is fetched by lane 0 inside the survivor path with VECTOR loads -- into registers the matrix-core scan's tile loop also
// uses, so that every LDS read of its exact path first waited for every outstanding memory operation of the wave (vmcnt(0)),
// the in-flight query-tile copies included.
__device__ __forceinline__ ScanExtra load_scan_extra(const ScanExtra *xp) {
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    static_assert(sizeof(ScanExtra) == 64, "one s_load_dwordx16");
    u32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(xp) : "memory");
    auto ptr = [&](int i) { return ((unsigned long long)v[i + 1] << 32) | v[i]; };
    ScanExtra e;
    e.arena_places = reinterpret_cast<uint2 *>(ptr(0));
    e.reserved = reinterpret_cast<const uint32_t *>(ptr(2));
    e.arena_recs = reinterpret_cast<SurvRec *>(ptr(4));
    e.arena_runs = reinterpret_cast<uint4 *>(ptr(6));
    e.arena_cur = reinterpret_cast<unsigned long long *>(ptr(8));
    e.arena_fail = reinterpret_cast<unsigned int *>(ptr(10));
    e.arena_sub = v[12], e.arena_rsub = v[13], e.arena_common = v[14], e.pad = v[15];
    return e;
}
__device__ __forceinline__ bool arena_reserve(const ScanExtra &a, uint32_t nrec, uint32_t nrun, uint32_t *rec_off, uint32_t *run_off) {
    const uint32_t shard = blockIdx.x & (RQ_ARENA_SHARDS - 1u);
    const unsigned long long o = atomicAdd(a.arena_cur + shard, ((unsigned long long)nrun << 32) | nrec);
    const uint32_t ab = (uint32_t)o, rb = (uint32_t)(o >> 32);
    if (ab + nrec <= a.arena_sub && rb + nrun <= a.arena_rsub) {
        *rec_off = shard * a.arena_sub + ab, *run_off = shard * a.arena_rsub + rb;
        return true;
    }
    atomicMin(a.arena_fail + shard, rb);  // the shard's valid runs end here (every later append fails as well)
    const unsigned long long oc = atomicAdd(a.arena_cur + RQ_ARENA_SHARDS + 2, ((unsigned long long)nrun << 32) | nrec);
    const uint32_t cb = (uint32_t)oc, crb = (uint32_t)(oc >> 32);
    if (cb + nrec <= a.arena_common && crb + nrun <= a.arena_common) {
        *rec_off = RQ_ARENA_SHARDS * a.arena_sub + cb, *run_off = RQ_ARENA_SHARDS * a.arena_rsub + crb;
        return true;
    }
    *reinterpret_cast<unsigned int *>(a.arena_cur + RQ_ARENA_SHARDS) = 1u;
    return false;
}
// (the scans record either into the uniform buffers or, in their ARENA instantiations, into the arena: segments are what
// the scatter pass and the consumers see)
__device__ __forceinline__ QSeg scan_seg(const ScanArgs &a) { return QSeg{nullptr, nullptr, a.cap}; }
struct ScanPtrs {   // host-side bundle only
    const uint32_t *codes;        // n * 2W dwords (x_binary_vec, src/rabitq.rs:66)
    const float4 *factors;        // n (src/rabitq.rs:67): x=factor_ip y=factor_ppc z=error_bound w=cds
    const uint32_t *grp_start;    // cluster-major: k+1 offsets into the record array
    const uint32_t *grp_cnt;      // cluster-major: records per list
    const uint32_t *offsets;      // k+1 list offsets of the index
    const uint32_t *recs;         // per-stage work records (stage_fill_kernel)
    SurvRec *surv;                // per query `cap` records
    RunRec *runs;                 // per query `cap` run descriptors
    unsigned long long *surv_cnt; // per query: low 32 bits = records, high 32 bits = runs
    unsigned long long *stat;     // matrix-core scan: 64 x {sub-tile steps, exact-path steps}
    const uint4 *tile_table;      // {list, first position of the tile in the list, list begin, list length} (use_table)
    const float4 *list_uref;      // additive gate: U0 per list (index)
    const float4 *grp_vref;       // additive gate: V0, DV per list (stage; group_vrange_kernel)
};
#define SCAN_PARAMS                                                                                  \
    const uint32_t *__restrict__ codes, const float4 *__restrict__ factors,                          \
        const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ grp_start,                \
        const uint32_t *__restrict__ recs, SurvRec *__restrict__ surv, RunRec *__restrict__ runs,    \
        unsigned long long *__restrict__ surv_cnt, const uint4 *__restrict__ tile_table, const ScanArgs a

// 8 code bits -> 8 nibbles (bit i -> nibble i), so that sum_j bit_j * q_j becomes v_dot8_u32_u4
__device__ __forceinline__ uint32_t spread8(uint32_t b) {
    uint32_t x = b & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x;
}

__device__ __forceinline__ float rough_distance(uint32_t s, const float4 &f, float lower, float delta,
                                                float sumq, float ycd, float ycd_sqrt) {
    float sf = (float)s;
    float t = f.w + ycd;                    // center_distance_square + y_c_distance_square
    t = t + lower * f.y;                    // + lower_bound * factor_ppc
    float u = (2.0f * sf - sumq) * f.x;     // (2 * dot - scalar_sum) * factor_ip
    t = t + u * delta;                      //   ... * delta
    return t - f.z * ycd_sqrt;              // - error_bound * dist_sqrt
}


#include "kernels_scan_decl.h"
