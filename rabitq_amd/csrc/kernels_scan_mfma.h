// kernels_scan_mfma.h -- the matrix-core scan kernel (definition; geometry helpers and the declaration: kernels_scan_decl.h)
#pragma once
#include "kernels_scan_common.h"

// ------------------------------------------------------------------------------------------------
// The same scan on the matrix cores, for stages where many queries share each list.
//
// sum_p popcount(code & plane_p) << p == sum_j bit_j(code) * q_j.  In a batch the VALU form above is
// issue-bound (v_dot8_u32_u4 delivers ~1 dimension per lane-cycle).  Every q_j in 0..15 is exactly
// q_j/2 in fp6 e2m3 and a code bit is exactly 1.0, so v_mfma_f32_32x32x64_f8f6f4 (A = e2m3, B = e2m3)
// gives s/2 EXACTLY in f32 for 32 queries x 32 candidates x 64 dimensions in 32 cycles: twice the
// rate of the i8 form, four times bf16.  Roles: A = 32 queries (their fp6 images, from the stage
// records through LDS), B = 32 candidates (code bits expanded to fp6 ONCE per block through a
// 256-entry byte -> 48-bit table in LDS, then resident in VGPRs for every query tile of the list),
// D lane map: column = candidate (lane & 31), the 16 registers x 2 half-waves = the 32 query rows.
// So for one accumulator register a __ballot gives, per half-wave, one query's gate over 32
// CONSECUTIVE list positions: the same run protocol as the VALU kernel.
//
// The gate itself is hoisted out of f32: rough < thr  <=>  s > S*(query, candidate), and the 32x32
// tile of S*/2 is ONE v_mfma_f32_32x32x16_bf16 (stage_fill_kernel explains the split and the margin),
// so the hot epilogue is 16 compares.  Only accumulator registers with a flagged lane evaluate the
// reference's f32 expression (src/rabitq.rs:352-363), and only its verdict is used.
//
// block = 4 waves; wave w owns NT sub-tiles of 32 positions: first + w*32*NT + t*32 + (lane&31).
// LDS: table 2 KiB + 2 x (32 query operands, row stride 12W+2 dwords: conflict-free ds_read_b64)
// + 2 x 20 x 32 transposed record tails.
// ------------------------------------------------------------------------------------------------
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int v16i32 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int imax3(int a, int b, int c) {
    const int m = a > b ? a : b;
    return m > c ? m : c;
}
typedef int v4i32 __attribute__((ext_vector_type(4)));
#ifndef RQ_F32X16_DEFINED
#define RQ_F32X16_DEFINED
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

// One 16-byte-per-lane LDS-DMA copy: lane l's 16 bytes at gsrc land at LDS byte address lds_dst + 16 l
// (lds_dst wave-uniform).  Issued from an asm statement so that the compiler's own s_waitcnt bookkeeping
// does not drain it early; completion is counted by hand (s_waitcnt vmcnt(N)) before the barrier that
// precedes the first read.
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// ADD (additive gate; dim <= 128, uniform survivor buffers): the rank-5 threshold S*(c, q) = sum_r u'_c[r] v'_q[r] + v4_q is replaced
// by the additive lower bound
//     S*(c, q) >= B_q + G_c,   B_q = sum_r U0[r] v'_q[r] + v4_q,   G_c = sum_r (d_c[r] V0[r] - |d_c[r]| DV[r]),   d_c = u'_c - U0
// with U0 a per-LIST reference of u' (mean over the list, list_uref_kernel: one float4 per list, part of the index) and
// V0 / DV the centre / half-range of v' over the pairs that probe the list in THIS stage (group_vrange_kernel):
// u'v' = U0 v' + d V0 + d (v' - V0) and |v' - V0| <= DV.  B_q travels with the query's record as the accumulator's start value
// C_q = -(B_q - margin) / 2 (stage_fill_kernel; the C operand of the first fp6 MFMA), G_c is a per-candidate constant of the block,
// and the hot gate is "max over the 16 accumulator registers > H_c = G_c / 2" -- the bf16 threshold MFMA (a third of the matrix
// cycles at dim 128, half at dim 64) is gone.  Measured looseness on the benchmark mixture (scripts/exp/additive_gate_sim.py):
// the bound sits 10-17 below S* where the cells' s sits 166 +- 26 below it: 1e-4 .. 1e-3 of the sub-tile steps are flagged
// (exact threshold: < 1e-5), each of which is then decided by the exact f32 expression exactly as before.
// The candidate operand of this form is fp4 (e2m1: a code bit is 1.0 = 0b0010; same MFMA rate as fp6 x fp6, 4 instead of 6
// registers per 32 dimensions), which pays for the 16 registers of C.
template <int W, int NT, bool ARENA, bool ADD>
__global__ __launch_bounds__((64 * scan_mfma_waves<W, ARENA>()), scan_mfma_blocks_per_cu<W>() /* = waves per SIMD: hipcc's second bound counts waves per execution unit */) void scan_mfma_kernel(const uint32_t *__restrict__ codes,
                                                           const float4 *__restrict__ factors,
                                                           const uint32_t *__restrict__ offsets,
                                                           const uint32_t *__restrict__ grp_start,
                                                           const uint32_t *__restrict__ grp_cnt,
                                                           const uint32_t *__restrict__ recs,
                                                           SurvRec *__restrict__ surv, RunRec *__restrict__ runs,
                                                           unsigned long long *__restrict__ surv_cnt,
                                                           unsigned long long *__restrict__ stat /* [128]: sub-tile steps / exact-path steps, 64 pairs by block */,
                                                           const uint4 *__restrict__ tile_table,
                                                           const float4 *__restrict__ list_uref /* ADD: U0 per list */,
                                                           const float4 *__restrict__ grp_vref /* ADD: V0, DV per list (two float4) */,
                                                           const ScanArgs a) {
    // (an ADD + ARENA instantiation was built and measured in round 4: on the hard distribution -- overlapping clusters, hub lists with
    // wide v' ranges -- the additive bound flags 95 % of the steps, 45 -> 131 ms per step: the arena stages keep the bf16 threshold)
    static_assert(!(ADD && ARENA), "the additive gate is built for the uniform survivor buffers only");
    constexpr uint32_t OPDW = 12 * W;            // operand dwords per record: dim fp6 fields
    constexpr uint32_t OPLD = rq_img_opld(OPDW, ADD);  // row stride (dwords) of the operand image: conflict-free ds_read_b64 (ADD at dim 128: ds_read_b128)
    constexpr uint32_t IMG_OP = 32 * OPLD;       // a query tile image: 32 operand rows ...
    constexpr uint32_t TAILD = ADD ? RQ_RECA_TAIL : RQ_REC_TAIL;
    constexpr uint32_t IMG = scan_mfma_img_dwords<W, ARENA, ADD>();  // ... + the 32 record tails (+ the 32 start values)
    constexpr uint32_t IMG_C = IMG_OP + 32 * TAILD;  // ADD: the 32 accumulator start values
    constexpr uint32_t NW = scan_mfma_waves<W, ARENA>();  // waves per block
    constexpr uint32_t WQ4 = IMG / (4 * NW);     // 16-byte pieces each wave copies (its share of the image)
    constexpr uint32_t NI = (WQ4 + 63) / 64;     // LDS-DMA instructions per wave per tile
    static_assert(IMG % (4 * NW) == 0, "tile image must split into NW 16-byte-aligned shares");
    constexpr uint32_t TILE = 32 * NW * NT;
    // candidate operand: fp4 e2m1 fields (a code bit = 1.0 = 0b0010), 4 dwords per 32 dimensions, in EVERY instantiation since round 5
    // (rounds <= 4: fp6 e2m3, 6 dwords, outside the additive form) -- same matrix rate, same exact products, a third fewer resident
    // registers: 12 per wave at dim 128 / three sub-tiles, 48 at dim 768
    constexpr uint32_t BDW = 4;
    __shared__ __attribute__((aligned(16))) uint32_t lut[256];
    extern __shared__ __attribute__((aligned(16))) uint32_t ring[];  // scan_mfma_ring_slots<W, ARENA>() x IMG dwords: query tiles in flight (LDS-DMA targets)
    __shared__ __attribute__((aligned(16))) float4 facL[TILE];      // the tile's factors, for the exact path
    // per-wave emit queue: survivors are parked here and written out in bulk (one atomic round trip per flush)
    constexpr uint32_t QE = 128, QR = 32;
    __shared__ uint32_t q_pos[NW][QE];
    __shared__ float q_rough[NW][QE];
    __shared__ uint32_t q_run[NW][QE];
    __shared__ uint32_t r_b[NW][QR], r_slot[NW][QR], r_pos[NW][QR], r_cnt[NW][QR], r_off[NW][QR], r_base[NW][QR];

    uint32_t g, first, list_begin, list_len;
    if (a.use_table) {  // one block per existing (list, tile); the list's bounds come with the entry
        const uint4 d = tile_table[a.group_base + blockIdx.x];
        g = d.x, first = d.y, list_begin = d.z, list_len = d.w;
    } else {
        const uint32_t gl = blockIdx.x / a.tiles_per_group;
        g = a.group_base + gl;
        first = (a.tile_base + (blockIdx.x - gl * a.tiles_per_group)) * TILE;
        list_begin = offsets[g], list_len = offsets[g + 1] - list_begin;
    }
    // the group's records (cluster-major only)
    const uint32_t pb = grp_start[g], cnt = grp_cnt[g];
    if (cnt == 0) return;
    if (first >= list_len) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ntiles = (cnt + 31) / 32;
#ifdef RQ_DEV_ABLATIONS
    const unsigned long long tm_begin = (a.dbg & 256u) ? __builtin_readcyclecounter() : 0ull;
#endif

    // everything the block needs from memory is requested up front: this lane's candidates (its half of
    // every code word + factors) and the first two query tiles
    uint32_t craw[NT][W];  // lane half h holds dims 64m + 32h .. +31 of candidate j of a sub-tile
    float4 fac0[NT];
    uint32_t lpos[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        lpos[t] = first + wave * (32 * NT) + t * 32 + j;
        const uint32_t pos = list_begin + (lpos[t] < list_len ? lpos[t] : 0);
        const uint32_t *cp = codes + (uint64_t)pos * (2 * W);
        fac0[t] = factors[pos];
#pragma unroll
        for (int m = 0; m < W; ++m) craw[t][m] = cp[2 * m + h];
    }
    float4 u0r = {0, 0, 0, 0}, v0r = {0, 0, 0, 0}, dvr = {0, 0, 0, 0};
    if constexpr (ADD) u0r = list_uref[g], v0r = grp_vref[2 * g], dvr = grp_vref[2 * g + 1];
    const uint32_t ring0 = lds_addr(&ring[0]);
    auto dma_tile = [&](uint32_t qt, uint32_t slot) {  // this wave's quarter of query tile qt -> ring[slot]
        if (RQ_DBG(a, 2u) && qt >= scan_mfma_ring_slots<W, ARENA>()) return;  // ablation: no re-staging (tiles re-use stale slots)
        const uint32_t *src = recs + ((uint64_t)(pb >> 5) + qt) * IMG + wave * (IMG / NW);
        const uint32_t dst = ring0 + (slot * IMG + wave * (IMG / NW)) * 4;
#pragma unroll
        for (uint32_t i = 0; i < NI; ++i) {
            const uint32_t q4 = i * 64 + lane;
            if (q4 < WQ4) glds16(src + q4 * 4, dst + i * 1024);  // same active lanes in every wave: NI issues each
        }
    };
    constexpr uint32_t QPB = scan_mfma_tiles_per_barrier<W>();
    if constexpr (QPB == 1) {
#pragma unroll
        for (uint32_t i = 0; i + 1 < scan_mfma_ring_slots<W, ARENA>(); ++i)
            if (i < ntiles) dma_tile(i, i);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < scan_mfma_periods_ahead<W, ARENA>() * QPB; ++i)
            if (i < ntiles) dma_tile(i, i);
    }
    if (tid < 256) {
        const uint32_t b = tid;
        {  // byte -> 8 fp4 fields (bit e -> 1.0 = 0b0010 at bits 4e .. 4e+3)
            uint32_t f = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) f |= ((b >> e) & 1u) << (4 * e + 1);
            lut[b] = f;
        }
    }

    uint32_t ub[ADD ? 1 : NT][4];   // B operand of the threshold MFMA: 8 bf16 per lane (slots 8h .. 8h+7)
    float hc[ADD ? NT : 1];         // ADD: the candidate's side of the gate, H_c = G_c / 2 (-inf: always flagged)
    bool forced = false;  // candidates whose factors do not admit the integer-threshold form
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // u'_c = (1, cds, ppc, eb) / factor_ip, 1   (factor_ip < 0 for every regular vector, rabitq.rs:227).
        // Lane half 0 carries u'[0], u'[1], half 1 carries u'[2], u'[3]; an approximate reciprocal is enough
        // (its 1 ulp is far inside the margin the bf16 split already needs).
        const float rf = __builtin_amdgcn_rcpf(fac0[t].x);
        const float mag = (1.0f + fabsf(fac0[t].w) + fabsf(fac0[t].y) + fabsf(fac0[t].z)) * fabsf(rf);
        const bool ok = fac0[t].x < 0.0f && mag < 1.0e37f;  // false for NaN / inf / factor_ip >= 0
        if constexpr (ADD) {
            const float u0 = rf, u1 = fac0[t].w * rf, u2 = fac0[t].y * rf, u3 = fac0[t].z * rf;
            const float d0 = u0 - u0r.x, d1 = u1 - u0r.y, d2 = u2, d3 = u3 - u0r.w;
            float gc = d0 * v0r.x - fabsf(d0) * dvr.x;
            gc += d1 * v0r.y - fabsf(d1) * dvr.y;
            gc += d2 * v0r.z - fabsf(d2) * dvr.z;
            gc += d3 * v0r.w - fabsf(d3) * dvr.w;
            // the f32 roundings of u', d and of the sums above, and of v' inside V0 / DV (2^-20 of the magnitudes involved;
            // the query's own margin covers its side, stage_fill_kernel)
            const float slop = ((fabsf(u0) + fabsf(u0r.x)) * (fabsf(v0r.x) + dvr.x) + (fabsf(u1) + fabsf(u0r.y)) * (fabsf(v0r.y) + dvr.y) +
                                fabsf(u2) * (fabsf(v0r.z) + dvr.z) + (fabsf(u3) + fabsf(u0r.w)) * (fabsf(v0r.w) + dvr.w)) *
                               (1.0f / 1048576.0f);
            const float hv = 0.5f * (gc - slop);
            const bool fin = fabsf(hv) < 1.0e37f;  // false for NaN / inf
            hc[t] = ok && fin ? hv : -__builtin_inff();
        } else {
            const float x0 = (h ? fac0[t].y : 1.0f) * rf, x1 = (h ? fac0[t].z : fac0[t].w) * rf;
            f32x2 xs = {x0, x1};
            const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(xs, bf16x2));  // xh0 | xh1 << 16 (RNE)
            f32x2 res = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u)};
            const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(res, bf16x2));  // xl0 | xl1 << 16
            const uint32_t one = 0x3F80u;
            // slots of this half: xh0 xl0 xh0 xh1 xl1 xh1 1 (1 | 0)
            ub[t][0] = ok ? ((hi & 0xFFFFu) | (lo << 16)) : 0u;
            ub[t][1] = ok ? hi : 0u;
            ub[t][2] = ok ? ((lo >> 16) | (hi & 0xFFFF0000u)) : 0u;
            ub[t][3] = ok ? (h ? one : (one | (one << 16))) : 0u;
            if (!ok) forced = true;
        }
    }
    // bf16-threshold form: forced candidates as a scalar, together with the developer switch: the hot loop tests one SGPR
    // instead of rebuilding the condition.  The hot test "some cell positive" is ONE compare against a wave-uniform bound:
    // 1 normally (a positive float is an int32 >= 1), INT_MIN when some candidate of the wave is forced (always true),
    // INT_MAX under the exact-off ablation.  (Additive form: a forced candidate's H_c is -inf, nothing else is needed.)
    uint32_t force_any = 0;
    int gate_min = 1;
    if constexpr (!ADD) {
        const uint64_t forcemask = __ballot(forced);
        force_any = __builtin_amdgcn_readfirstlane(forcemask != 0ull ? 1u : 0u);
        const uint32_t exact_off = __builtin_amdgcn_readfirstlane(RQ_DBG(a, 64u) ? 1u : 0u);
        gate_min = (int)__builtin_amdgcn_readfirstlane(exact_off ? 0x7FFFFFFFu : (force_any ? 0x80000000u : 1u));
    } else {
#ifdef RQ_DEV_ABLATIONS
        if (a.dbg & 64u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) hc[t] = __builtin_inff();  // exact path off (timing ablation)
        }
#endif
    }
    if (h == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) facL[lpos[t] - first] = fac0[t];
    }

    // ---- the emit queue of this wave ----
    uint32_t nE = 0, nR = 0;  // wave-uniform fill levels
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // LDS is in-order per wave
        const QSeg seg = scan_seg(a);
        if constexpr (ARENA) {  // arena mode: count per query, ONE reservation for the whole queue in this block's shard
            const ScanExtra xe = load_scan_extra(a.x);  // (one scalar load: see there)
            unsigned long long place = 0;  // records | runs << 32 of the query before this run: its place in the query's segment
            if (lane < nR) place = atomicAdd(surv_cnt + r_b[wave][lane], (1ull << 32) | r_cnt[wave][lane]);
            uint32_t off0 = 0xFFFFFFFFu, roff = 0;
            if (lane == 0 && nR && !arena_reserve(xe, nE, nR, &off0, &roff)) off0 = 0xFFFFFFFFu;
            off0 = __builtin_amdgcn_readfirstlane(off0), roff = __builtin_amdgcn_readfirstlane(roff);
            if (off0 != 0xFFFFFFFFu) {
                SurvRec *arecs = xe.arena_recs;
                uint4 *aruns = xe.arena_runs;
                if (lane < nR) {
                    aruns[roff + lane] =
                        make_uint4(r_pos[wave][lane], r_slot[wave][lane] | (r_cnt[wave][lane] << 16), r_b[wave][lane], off0 + r_off[wave][lane]);
                    xe.arena_places[roff + lane] = make_uint2((uint32_t)place, (uint32_t)(place >> 32));
                }
                for (uint32_t e = lane; e < nE; e += 64) {
                    SurvRec sr;
                    sr.pos = q_pos[wave][e], sr.slot = r_slot[wave][q_run[wave][e]], sr.rough = q_rough[wave][e], sr.accurate = 0.0f;
                    arecs[off0 + e] = sr;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            nE = 0, nR = 0;
            return;
        }
        if constexpr (ARENA) return;  // (unreachable)
        if (lane < nR) {  // one lane per run: all reservations in flight together
            const uint32_t rb = r_b[wave][lane], rc = r_cnt[wave][lane];
            const unsigned long long old = atomicAdd(surv_cnt + rb, (1ull << 32) | rc);
            const uint32_t base = (uint32_t)old, rbase = (uint32_t)(old >> 32);
            r_base[wave][lane] = base;
            if (rbase < seg.capof(rb)) {
                RunRec rr;
                rr.pos = r_pos[wave][lane];
                rr.slot = r_slot[wave][lane];
                rr.base = base;
                rr.cnt = rc;
                runs[seg.at(rb) + rbase] = rr;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (uint32_t e = lane; e < nE; e += 64) {
            const uint32_t r = q_run[wave][e];
            const uint32_t at = r_base[wave][r] + (e - r_off[wave][r]);
            const uint32_t rb = r_b[wave][r];
            if (at < seg.capof(rb)) {
                SurvRec sr;
                sr.pos = q_pos[wave][e];
                sr.slot = r_slot[wave][r];
                sr.rough = q_rough[wave][e];
                sr.accurate = 0.0f;
                surv[seg.at(rb) + at] = sr;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // nothing of the flush may stay outstanding: the counted waits below assume only LDS-DMA is in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        nE = 0, nR = 0;
    };

    __syncthreads();  // table visible (the LDS-DMA is invisible to this barrier's fence)

    // B: this lane's code bits as fp6 (fp4: ADD) fields, BDW dwords per 32 dimensions, resident for the whole block
    uint32_t bexp[NT][W][BDW];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int m = 0; m < W; ++m) {
            const uint32_t c = craw[t][m];
            bexp[t][m][0] = lut[c & 0xFFu], bexp[t][m][1] = lut[(c >> 8) & 0xFFu];
            bexp[t][m][2] = lut[(c >> 16) & 0xFFu], bexp[t][m][3] = lut[c >> 24];
        }

    // always-on statistic (results unchanged, nothing in the hot loop): 32x32 sub-tile steps taken = NT per tile, and how many of
    // them took the exact path.  The host reads the totals with the pass's other counters and drops the additive gate for an
    // index on which it flags too much.
    uint32_t n_flag = 0;
#ifdef RQ_DEV_ABLATIONS
    // developer hook (dbg & 256): cycles of the block's start-up, of the waits at the top of the tile loop and of the
    // tile bodies, summed over blocks into stat[128..131) (+ block count): where a wave's lifetime goes
    const uint32_t time_stat = __builtin_amdgcn_readfirstlane((a.dbg & 256u) ? 1u : 0u);
    // timing ablation (dbg & 8192; results are wrong): the tile loop without its block barriers -- what the waves of a block
    // lose by waiting for each other
    const uint32_t no_barrier = __builtin_amdgcn_readfirstlane((a.dbg & 8192u) ? 1u : 0u);
    unsigned long long tm_wait = 0, tm_body = 0, tm_mark = 0, tm_startup = 0, tm_exact = 0, tm_flush = 0;
    uint32_t n_regs = 0, n_flush = 0, n_greg = 0;
    if (time_stat) {
        tm_mark = __builtin_readcyclecounter();
        tm_startup = tm_mark - tm_begin;
    }
#else
    constexpr uint32_t no_barrier = 0;
#endif
    uint32_t slot = 0;  // ring slot of query tile qt
    for (uint32_t qt = 0; qt < (RQ_DBG(a, 4u) ? 0u : ntiles); ++qt) {
        if constexpr (QPB == 1) {  // SLOTS slots, one barrier per tile, PD = SLOTS - 1 tiles in flight
            constexpr uint32_t SLOTS = scan_mfma_ring_slots<W, ARENA>(), PD = SLOTS - 1;
            // tile qt has landed once only the copies of the (up to PD - 1) later tiles are still in flight (in-order counter)
            const uint32_t later = ntiles - 1 - qt < PD - 1 ? ntiles - 1 - qt : PD - 1;  // wave-uniform
            if (later == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");
            static_assert(PD <= 4 && 3 * NI < 64, "vmcnt immediates are spelled out for up to three later tiles");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!no_barrier) __builtin_amdgcn_s_barrier();  // every wave's quarter of tile qt is in; everyone is done with tile qt-1
            if (qt + PD < ntiles) dma_tile(qt + PD, slot == 0 ? SLOTS - 1 : slot - 1);  // into the slot tile qt-1 occupied
        } else if (qt % QPB == 0) {  // (1 + AHEAD) * QPB slots, one barrier per QPB tiles: tiles qt .. qt+QPB-1 were requested
                                     // AHEAD barriers ago, the copies of the AHEAD - 1 periods after them may still be in
                                     // flight (in-order counter), and the QPB tiles AHEAD periods on go out now
            constexpr uint32_t AHEAD = scan_mfma_periods_ahead<W, ARENA>();
            static_assert(QPB == 2 && AHEAD <= 2 && 2 * NI < 64, "the vmcnt immediates below are spelled out for two tiles per barrier, two periods");
            const uint32_t later = AHEAD < 2 || ntiles - qt <= QPB ? 0u : (ntiles - qt - QPB < QPB ? ntiles - qt - QPB : QPB);  // wave-uniform
            if (later == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!no_barrier) __builtin_amdgcn_s_barrier();
#pragma unroll
            for (uint32_t i = 0; i < QPB; ++i)
                if (qt + AHEAD * QPB + i < ntiles) dma_tile(qt + AHEAD * QPB + i, (slot + AHEAD * QPB + i) % ((1 + AHEAD) * QPB));
        }
#ifdef RQ_DEV_ABLATIONS
        if (time_stat) {
            const unsigned long long now = __builtin_readcyclecounter();
            tm_wait += now - tm_mark;
            tm_mark = now;
        }
#endif
        const uint32_t *img = ring + slot * IMG;
        const uint32_t nvalid = cnt - 32 * qt;  // rows >= nvalid of the last tile are stale memory: masked here

        // A: query row j (= lane & 31), dims 64m + 32h .. +31 as fp6.  Rows >= nvalid hold stale bytes: harmless,
        // every fp6 pattern is a finite number and such a row's accumulator starts at -inf (below)
        constexpr bool STREAM_A = scan_mfma_stream_a<W>();
        uint32_t aop[STREAM_A ? 1 : W][6];
        auto load_a = [&](int m) {  // the 6 dwords of slab m of this lane's query row
            v8i32 av = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 6; e += 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(&img[j * OPLD + 6 * W * h + 6 * m + e]);
                av[e] = (int)v.x, av[e + 1] = (int)v.y;
            }
            return av;
        };
        if constexpr (ADD && W == 2) {  // the lane's 12 dwords in three 16-byte reads
#pragma unroll
            for (int e = 0; e < 12; e += 4) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&img[j * OPLD + 12 * h + e]);
                (&aop[0][0])[e] = v.x, (&aop[0][0])[e + 1] = v.y, (&aop[0][0])[e + 2] = v.z, (&aop[0][0])[e + 3] = v.w;
            }
        } else if constexpr (!STREAM_A) {
#pragma unroll
            for (int m = 0; m < W; ++m) {
                const v8i32 av = load_a(m);
#pragma unroll
                for (int e = 0; e < 6; ++e) aop[m][e] = (uint32_t)av[e];
            }
        }
        auto get_a = [&](int m) {
            if constexpr (STREAM_A) {
                return load_a(m);
            } else {
                const v8i32 av = {(int)aop[m][0], (int)aop[m][1], (int)aop[m][2], (int)aop[m][3], (int)aop[m][4], (int)aop[m][5], 0, 0};
                return av;
            }
        };
        auto get_b = [&](int t, int m) {
            const v8i32 bv = {(int)bexp[t][m][0], (int)bexp[t][m][1], (int)bexp[t][m][2], (int)bexp[t][m][3], 0, 0, 0, 0};
            return bv;
        };
        // one 32 x 32 x 64 product block on top of c: A fp6 (e2m3) x B fp6 / fp4 (e2m1), exact in f32
        auto mm = [&](const v8i32 av, const v8i32 bv, const f32x16 c) {
            return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 2 /*A e2m3*/, 4 /*B e2m1*/, 0, 0, 0, 0);
        };
        // bf16 form: A operand of the threshold MFMA, slots 8h .. 8h+7 of query row j
        // (rows past the list's last query carry the "no query" operand written by group_scan_kernel: -S* = -inf)
        v4i32 ua = {0, 0, 0, 0};
        // additive form: the accumulator tile's start values C_q (register gq of lane half h = query row (gq & 3) + 8 (gq >> 2) + 4h;
        // -inf for rows past the list's last query, +inf for a query whose scales do not admit the integer form: always flagged)
        f32x16 cinit = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (ADD) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 cv = *reinterpret_cast<const float4 *>(&img[IMG_C + 8 * g4 + 4 * h]);
                cinit[4 * g4] = cv.x, cinit[4 * g4 + 1] = cv.y, cinit[4 * g4 + 2] = cv.z, cinit[4 * g4 + 3] = cv.w;
            }
        } else {
            ua = *reinterpret_cast<const v4i32 *>(&img[IMG_OP + j * TAILD + RQ_REC_V0 + 4 * h]);
        }
        auto tail = [&](uint32_t f, uint32_t row) { return img[IMG_OP + row * TAILD + f]; };

        // accumulator tiles = -S*/2 (query row, candidate col) + s/2, all on the matrix pipe
        // Wide vectors: slab-outer, so that the NT accumulation chains interleave on the matrix pipe (a chain of dependent
        // MFMAs alone issues at ~44 cycles per instruction instead of 32) and one fragment load (when A is streamed)
        // feeds every sub-tile's accumulator
        constexpr bool SLAB_OUTER = W > 4;
        f32x16 accs[SLAB_OUTER ? NT : 1];
        if constexpr (SLAB_OUTER) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const v4i32 ubv = {(int)ub[t][0], (int)ub[t][1], (int)ub[t][2], (int)ub[t][3]};
                const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                accs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ubv), z, 0, 0, 0);
            }
            if constexpr (STREAM_A && RQ_A_RING > 1) {
                // A fragments RQ_A_RING slabs ahead of their use: left to itself the compiler requests a fragment one slab (two matrix
                // instructions, 64 cycles) before it is needed -- less than an LDS round trip under load -- and these instantiations
                // hold two waves per SIMD (SQ counters at dim 768: matrix pipe 0.475 busy, 0.46 of the wave-cycles waiting on an
                // instruction's operands).  The scheduling barriers pin the order written here; the waits land at the uses.
                constexpr int RING = RQ_A_RING < W ? RQ_A_RING : W;
                v8i32 af[RING];
#pragma unroll
                for (int m = 0; m < RING; ++m) af[m] = load_a(m);
#pragma unroll
                for (int m = 0; m < W; ++m) {
                    __builtin_amdgcn_sched_barrier(0);
                    const v8i32 av = af[m % RING];
#pragma unroll
                    for (int t = 0; t < NT; ++t) accs[t] = mm(av, get_b(t, m), accs[t]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m + RING < W) af[m % RING] = load_a(m + RING);
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int m = 0; m < W; ++m) {
                    const v8i32 av = get_a(m);
#pragma unroll
                    for (int t = 0; t < NT; ++t) accs[t] = mm(av, get_b(t, m), accs[t]);
                }
            }
        }
        // the cold path of one 32 x 32 step whose gate fired: the accumulator registers with a flagged lane, s itself for them, the
        // reference's f32 expression for those registers only, survivors into the wave's emit queue (t is a constant after unrolling)
        auto cold_path = [&](const int t, const f32x16 acc) __attribute__((always_inline)) {
            ++n_flag;
            // nothing of the cold branch may be scheduled ahead of it: hoisted into the tile loop its recomputation and LDS reads
            // cost the wide instantiations 34 spilled registers (dim 768: 21.6 -> 28.4 ms per launch)
            __builtin_amdgcn_sched_barrier(0);
#ifdef RQ_DEV_ABLATIONS
            const unsigned long long tx0 = time_stat ? __builtin_readcyclecounter() : 0ull;
#endif
            // lives inside this branch so that the common path carries no state of it (not even a zeroed tile)
            uint32_t gmask = force_any ? 0xFFFFu : 0u;  // accumulator registers with at least one flagged lane
            f32x16 sc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int gq = 0; gq < 16; ++gq) {
                if constexpr (ADD) gmask |= (__ballot(acc[gq] > hc[t]) != 0ull ? 1u : 0u) << gq;
                else gmask |= (__ballot(acc[gq] > 0.0f) != 0ull ? 1u : 0u) << gq;
            }
            // the flagged cells need s itself: the same products again on a clean accumulator (exact)
#pragma unroll
            for (int m = 0; m < W; ++m) sc = mm(get_a(m), get_b(t, m), sc);
            if (RQ_DBG(a, 1u)) gmask = 0;
#ifdef RQ_DEV_ABLATIONS
            if (time_stat) n_greg += (uint32_t)__popc(gmask);
#endif
            // exact evaluation + emit, for the flagged registers only
            const float4 fc = facL[lpos[t] - first];
            while (gmask) {
                const uint32_t gq = (uint32_t)__builtin_ctz(gmask);
                gmask &= gmask - 1;
                const uint32_t row = (gq & 3) + 8 * (gq >> 2) + 4 * h;
                const float sf = 2.0f * sc[gq];  // wave-uniform register index
                // the row's scalars in two 16-byte LDS reads: lower delta sumq ycd | ycd_sqrt thr lo hi
                const uint4 ta = *reinterpret_cast<const uint4 *>(&img[IMG_OP + row * TAILD]);
                const uint4 tb = *reinterpret_cast<const uint4 *>(&img[IMG_OP + row * TAILD + 4]);
                static_assert(RQ_REC_LOWER == 0 && RQ_REC_DELTA == 1 && RQ_REC_SUMQ == 2 && RQ_REC_YCD == 3 && RQ_REC_YCD_SQRT == 4 &&
                                  RQ_REC_THR == 5 && RQ_REC_LO == 6 && RQ_REC_HI == 7, "tail layout read as two uint4");
                // the reference's expression, left to right (src/rabitq.rs:352-363)
                float tt = fc.w + __builtin_bit_cast(float, ta.w);
                tt = tt + __builtin_bit_cast(float, ta.x) * fc.y;
                const float u = (2.0f * sf - __builtin_bit_cast(float, ta.z)) * fc.x;
                tt = tt + u * __builtin_bit_cast(float, ta.y);
                const float rg = tt - fc.z * __builtin_bit_cast(float, tb.x);
                bool pass = rg < __builtin_bit_cast(float, tb.y);  // src/rerank.rs:84
                // a real query, and a list position inside its stage range
                pass = pass && row < nvalid && lpos[t] >= tb.z && lpos[t] < tb.w;
                const uint64_t m = __ballot(pass);
                if (m == 0) continue;
#ifdef RQ_DEV_ABLATIONS
                if (time_stat) ++n_regs;
#endif
                if (nE + 64 > QE || nR + 2 > QR) {
#ifdef RQ_DEV_ABLATIONS
                    const unsigned long long tf0 = time_stat ? __builtin_readcyclecounter() : 0ull;
#endif
                    flush();
#ifdef RQ_DEV_ABLATIONS
                    if (time_stat) tm_flush += __builtin_readcyclecounter() - tf0, ++n_flush;
#endif
                }
                // each half-wave is one run (one query x 32 consecutive positions); half 0 first
                const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
                const uint32_t c0 = (uint32_t)__popc(m0), c1 = (uint32_t)__popc(m1);
                const uint32_t myrun = nR + ((h && c0) ? 1u : 0u), myoff = nE + (h ? c0 : 0u);
                if (pass) {
                    const uint32_t e = myoff + (uint32_t)__popc((h ? m1 : m0) & ((1u << j) - 1u));
                    q_pos[wave][e] = list_begin + lpos[t];
                    q_rough[wave][e] = rg;
                    q_run[wave][e] = myrun;
                }
                if (j == 0 && (h ? c1 : c0)) {
                    r_b[wave][myrun] = tail(RQ_REC_ROW, row);
                    r_slot[wave][myrun] = tail(RQ_REC_SLOT, row);
                    r_pos[wave][myrun] = list_begin + first + wave * (32 * NT) + t * 32;
                    r_cnt[wave][myrun] = h ? c1 : c0;
                    r_off[wave][myrun] = myoff;
                }
                nE += c0 + c1;
                nR += (c0 ? 1u : 0u) + (c1 ? 1u : 0u);
            }
#ifdef RQ_DEV_ABLATIONS
            if (time_stat) tm_exact += __builtin_readcyclecounter() - tx0;
#endif
        };
        // Additive form, narrow vectors: ONE branch per query tile.  The NT steps run back to back, each leaving only its gate's
        // verdict in a scalar mask (v_cmp into an SGPR pair, s_or); the tile's cold path -- 7.8e-4 of the steps on the benchmark
        // mixture -- recomputes a step's accumulator tile (same instructions, same bits) before it looks at it.
        constexpr bool DEFER = ADD && !SLAB_OUTER && (RQ_GATE_DEFER != 0);
        if constexpr (DEFER) {
            uint64_t anyhot = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x16 acc = cinit;
#pragma unroll
                for (int m = 0; m < W; ++m) acc = mm(get_a(m), get_b(t, m), acc);
                float mx = hc[t];
#pragma unroll
                for (int gq = 0; gq < 16; gq += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, acc[gq]), acc[gq + 1]);
                anyhot |= __ballot(mx > hc[t]);
            }
            if (__builtin_expect(anyhot != 0ull, 0)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x16 acc = cinit;
#pragma unroll
                    for (int m = 0; m < W; ++m) acc = mm(get_a(m), get_b(t, m), acc);
                    float mx = hc[t];
#pragma unroll
                    for (int gq = 0; gq < 16; gq += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, acc[gq]), acc[gq + 1]);
                    if (__ballot(mx > hc[t]) != 0ull) cold_path(t, acc);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < (DEFER ? 0 : NT); ++t) {
            f32x16 acc;
            if constexpr (SLAB_OUTER) {
                acc = accs[t];
            } else if constexpr (ADD) {
                acc = cinit;
#pragma unroll
                for (int m = 0; m < W; ++m) acc = mm(get_a(m), get_b(t, m), acc);
            } else {
                const v4i32 ubv = {(int)ub[t][0], (int)ub[t][1], (int)ub[t][2], (int)ub[t][3]};
                const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ubv), z, 0, 0, 0);
#pragma unroll
                for (int m = 0; m < W; ++m) acc = mm(get_a(m), get_b(t, m), acc);
            }
            bool hot;
            if constexpr (ADD) {
                // hot path: does any of this lane's 16 cells exceed the candidate's H_c?  The chain starts at H_c itself, so the
                // reduction is 8 v_max3_f32 + 1 compare (no NaN can occur: C_q is finite or +-inf of one sign per row, the products
                // are finite)
                float mx = hc[t];
#pragma unroll
                for (int gq = 0; gq < 16; gq += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, acc[gq]), acc[gq + 1]);
                hot = __ballot(mx > hc[t]) != 0ull;
            } else {
                // hot path: is any of the 1024 (query, candidate) cells positive?  A float is positive iff its bit
                // pattern is a positive int32 (a NaN with a clear sign bit counts as positive: conservative), so the
                // reduction is 7 v_max3_i32 + 1 v_max_i32 + 1 compare, with no canonicalisation
                const v16i32 ai = __builtin_bit_cast(v16i32, acc);
                int mxi = imax3(ai[0], ai[1], ai[2]);
#pragma unroll
                for (int gq = 3; gq < 15; gq += 2) mxi = imax3(mxi, ai[gq], ai[gq + 1]);
                mxi = mxi > ai[15] ? mxi : ai[15];
                hot = __ballot(mxi >= gate_min) != 0ull;
            }
            // wave-uniform; everything below.  (Additive form: laid out off the hot path, +1.5 %; the wide instantiations spill with it.)
            if (ADD ? __builtin_expect(hot, 0) : hot) {
                cold_path(t, acc);
            }
        }
        slot = slot + 1 == scan_mfma_ring_slots<W, ARENA>() ? 0 : slot + 1;
#ifdef RQ_DEV_ABLATIONS
        if (time_stat) {
            const unsigned long long now = __builtin_readcyclecounter();
            tm_body += now - tm_mark;
            tm_mark = now;
        }
#endif
    }
#ifdef RQ_DEV_ABLATIONS
    if (time_stat && tid == 0) {
        atomicAdd(stat + 128, tm_startup), atomicAdd(stat + 129, tm_wait), atomicAdd(stat + 130, tm_body);
        atomicAdd(stat + 131, 1ull), atomicAdd(stat + 132, (unsigned long long)ntiles);
        atomicAdd(stat + 133, tm_exact), atomicAdd(stat + 134, tm_flush), atomicAdd(stat + 135, (unsigned long long)n_regs);
        atomicAdd(stat + 136, (unsigned long long)n_flush), atomicAdd(stat + 137, (unsigned long long)n_greg);
    }
#endif
    if (nE) flush();
    if (lane == 0) {  // 64 pairs of counters, by block: a single address would serialise a million atomics
        atomicAdd(stat + 2 * (blockIdx.x & 63u), (unsigned long long)(ntiles * NT));
        if (n_flag) atomicAdd(stat + 2 * (blockIdx.x & 63u) + 1, (unsigned long long)n_flag);
    }
}


