// kernels_scan_valu.h -- the VALU scan kernel (definition; kernels_query.h has the contract and the record layout)
#pragma once
#include "kernels_scan_common.h"

// The integer part: the asymmetric dot product sum_p popcount(code & plane_p) << p
// (src/utils.rs:113-135, src/simd.rs:326-384) equals sum_j bit_j(code) * q_j exactly.  Each lane
// expands its candidates' code bits to nibbles ONCE per block (amortised over every query of the
// group) and the per-query work is 8W chained v_dot8_u32_u4 (8 dimensions each, u32 accumulate,
// query operand in an SGPR) instead of 8W v_and + 8W v_bcnt + adds.
template <int W, int CPL, bool ARENA>
__global__ __launch_bounds__(256) void scan_kernel(SCAN_PARAMS) {
    constexpr uint32_t STRIDE = 8 * W + RQ_REC_TAIL;
    uint32_t g, first, list_begin = 0, list_len = 0;
    if (a.use_table) {  // cluster-major, one block per existing (list, tile)
        const uint4 d = tile_table[a.group_base + blockIdx.x];
        g = d.x, first = d.y, list_begin = d.z, list_len = d.w;
    } else {
        const uint32_t gl = blockIdx.x / a.tiles_per_group;
        g = a.group_base + gl;
        first = (a.tile_base + (blockIdx.x - gl * a.tiles_per_group)) * (256 * CPL);  // first list position of this tile
        // the list of this group (all its pairs share it); cluster-major: straight from the index, in the same
        // round trip as the group bounds
        if (a.cluster_major) {
            list_begin = offsets[g];
            list_len = offsets[g + 1] - list_begin;
        }
    }
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
    if (!a.cluster_major) list_begin = rec[8 * W + RQ_REC_LIST_BEGIN], list_len = rec[8 * W + RQ_REC_LIST_LEN];
    if (first >= list_len) return;
    if (!a.cluster_major && rec[8 * W + RQ_REC_LO] >= rec[8 * W + RQ_REC_HI]) return;  // pair not in this stage

    uint32_t xn[CPL][8 * W];  // nibble-expanded codes
    float4 fac[CPL];
    uint32_t pos[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        uint32_t local = first + c * 256 + threadIdx.x;
        pos[c] = list_begin + (local < list_len ? local : 0);
        const uint32_t *cp = codes + (uint64_t)pos[c] * (2 * W);
        uint32_t code[2 * W];
        if constexpr ((2 * W) % 4 == 0) {
#pragma unroll
            for (int i = 0; i < 2 * W; i += 4) {
                uint4 v = *reinterpret_cast<const uint4 *>(cp + i);
                code[i] = v.x, code[i + 1] = v.y, code[i + 2] = v.z, code[i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2 * W; i += 2) {
                uint2 v = *reinterpret_cast<const uint2 *>(cp + i);
                code[i] = v.x, code[i + 1] = v.y;
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * W; ++j)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) xn[c][4 * j + kq] = spread8(code[j] >> (8 * kq));
        fac[c] = factors[pos[c]];
    }

    // one (query, list) pair of the group against this block's candidates; qn = the query's 8W operand dwords,
    // t = the record's tail (both wave-uniform: a pointer the compiler turns into scalar loads, or SGPR tuples)
    auto score_pair = [&](const auto &qn, const auto &t) {
        // positions of this list that belong to the stage: [lo_p, hi_p) (wave-uniform)
        const uint32_t lo_p = t[RQ_REC_LO], hi_p = t[RQ_REC_HI];
        if (hi_p <= first || lo_p >= first + 256 * CPL) return;
        // (through a by-value parameter: __builtin_bit_cast applied directly to an element of an SGPR tuple reads element 0)
        auto f32_of = [](uint32_t v) { return __builtin_bit_cast(float, v); };
        const float lower = f32_of(t[RQ_REC_LOWER]), delta = f32_of(t[RQ_REC_DELTA]), sumq = f32_of(t[RQ_REC_SUMQ]),
                    ycd = f32_of(t[RQ_REC_YCD]), ycd_sqrt = f32_of(t[RQ_REC_YCD_SQRT]), thr = f32_of(t[RQ_REC_THR]);
        float rough[CPL];
        uint32_t sdot[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            uint32_t acc = 0;
#pragma unroll
            for (int m = 0; m < 8 * W; ++m) acc = __builtin_amdgcn_udot8(xn[c][m], qn[m], acc, false);
            sdot[c] = acc;
        }
        if constexpr (CPL == 2) {
            // both candidates of a lane in packed f32 (v_pk_*): one rounding per op, same order
            f32x2 sf = {(float)sdot[0], (float)sdot[1]};
            f32x2 cds = {fac[0].w, fac[1].w}, ppc = {fac[0].y, fac[1].y}, fip = {fac[0].x, fac[1].x},
                  eb = {fac[0].z, fac[1].z};
            f32x2 tt = cds + ycd;
            tt = tt + lower * ppc;
            f32x2 u = (2.0f * sf - sumq) * fip;
            tt = tt + u * delta;
            f32x2 r = tt - eb * ycd_sqrt;
            rough[0] = r.x, rough[1] = r.y;
        } else {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                rough[c] = rough_distance(sdot[c], fac[c], lower, delta, sumq, ycd, ycd_sqrt);
        }
        uint32_t total = 0;
        uint64_t m[CPL];
        if (lo_p <= first && first + 256 * CPL <= hi_p) {  // whole tile inside the stage (the common case)
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                m[c] = __ballot(rough[c] < thr);  // src/rerank.rs:84 gate
                total += (uint32_t)__popcll(m[c]);
            }
        } else {  // boundary tile: lanes of this wave hold positions P0 .. P0+63, the in-stage ones are a bit range
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const uint32_t P0 = first + c * 256 + wave * 64;
                const uint32_t ra = lo_p > P0 ? (lo_p - P0 < 64 ? lo_p - P0 : 64) : 0;
                const uint32_t rb = hi_p > P0 ? (hi_p - P0 < 64 ? hi_p - P0 : 64) : 0;
                const uint64_t below_b = rb >= 64 ? ~0ull : ((1ull << rb) - 1ull);
                const uint64_t below_a = ra >= 64 ? ~0ull : ((1ull << ra) - 1ull);
                m[c] = __ballot(rough[c] < thr) & below_b & ~below_a;
                total += (uint32_t)__popcll(m[c]);
            }
        }
        if (RQ_DBG(a, 1024u)) total = 0;  // timing ablation: no survivor is recorded (results are wrong)
        if (total) {  // wave-uniform: one 64-bit atomic reserves the records and the run descriptors
            const uint32_t b = t[RQ_REC_ROW], slot = t[RQ_REC_SLOT];
            uint32_t nruns = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) nruns += m[c] ? 1u : 0u;
            if constexpr (ARENA) {  // arena mode: count per query, append to this block's shard of the arena
                uint32_t off = 0xFFFFFFFFu, roff = 0;
                const ScanExtra xe = load_scan_extra(a.x);  // (one scalar load)
                unsigned long long place = 0;  // the query's records | runs << 32 before these runs: their place in its segment
                if (lane == 0) {
                    place = atomicAdd(surv_cnt + b, ((unsigned long long)nruns << 32) | total);
                    if (!arena_reserve(xe, total, nruns, &off, &roff)) off = 0xFFFFFFFFu;
                }
                off = __builtin_amdgcn_readfirstlane(off), roff = __builtin_amdgcn_readfirstlane(roff);
                if (off == 0xFFFFFFFFu) return;
                uint32_t pbase = (uint32_t)place, prun = (uint32_t)(place >> 32);  // (lane 0's values: it writes the descriptors)
                SurvRec *arecs = xe.arena_recs;
                uint4 *rdst = xe.arena_runs + roff;
                uint2 *pdst = xe.arena_places + roff;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const uint32_t cntc = (uint32_t)__popcll(m[c]);
                    if ((m[c] >> lane) & 1ull) {
                        SurvRec r;
                        r.pos = pos[c], r.slot = slot, r.rough = rough[c], r.accurate = 0.0f;
                        arecs[off + (uint32_t)__popcll(m[c] & ((1ull << lane) - 1ull))] = r;
                    }
                    if (cntc && lane == 0) {
                        *rdst++ = make_uint4(list_begin + first + c * 256 + (threadIdx.x & ~63u), slot | (cntc << 16), b, off);
                        *pdst++ = make_uint2(pbase, prun);
                        pbase += cntc, ++prun;
                    }
                    off += cntc;
                }
                return;
            }
            if constexpr (ARENA) return;  // (unreachable: keeps the direct path out of the arena instantiation)
            // Direct path.  The stages this kernel serves in a large batch are survivor-dense (nearly every 128-pair iteration
            // gets here), so the emission is kept as short as the scoring: everything wave-uniform (query row, buffer base, run
            // bases, counts) lives in scalar registers, the ranks come from v_mbcnt, and the run descriptors of both sub-tiles
            // leave in ONE store (lane c writes run c).
            unsigned long long old = 0;
            if (lane == 0) old = atomicAdd(surv_cnt + b, ((unsigned long long)nruns << 32) | total);
            uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)old);
            const uint32_t rbase = __builtin_amdgcn_readfirstlane((uint32_t)(old >> 32));
            const uint32_t qcap = a.cap;  // (uniform geometry: query b owns slots [b * cap, (b + 1) * cap))
            const uint64_t qat64 = (uint64_t)b * qcap;
            const uint64_t qat = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(qat64 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)qat64);
            SurvRec *out = surv + qat;
            uint32_t cn[CPL], cbase[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                cn[c] = (uint32_t)__popcll(m[c]);
                cbase[c] = base;
                const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[c] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[c], 0u));
                if (((m[c] >> lane) & 1ull) && at < qcap) {
                    SurvRec r;
                    r.pos = pos[c];
                    r.slot = slot;
                    r.rough = rough[c];
                    r.accurate = 0.0f;
                    out[at] = r;
                }
                base += cn[c];
            }
            if (lane < CPL) {  // lane c: the descriptor of sub-tile c's run -- appended (sorted later), or straight into its cell of the dense directory
                uint32_t mycnt = cn[0], mybase = cbase[0], myr = rbase;
#pragma unroll
                for (int c = 1; c < CPL; ++c)
                    if (lane == (uint32_t)c) mycnt = cn[c], mybase = cbase[c], myr = rbase + (cn[c - 1] ? 1u : 0u);
                static_assert(CPL <= 2, "run slot of sub-tile c = rbase + non-empty sub-tiles before it");
                const uint32_t p0 = first + lane * 256 + (threadIdx.x & ~63u);
                const uint32_t dcell = a.dense_dir ? t[RQ_REC_CELL0] + (p0 >> 6) : myr;
                if (mycnt && dcell < qcap) {
                    RunRec rr;
                    rr.pos = list_begin + p0;
                    rr.slot = slot;
                    rr.base = mybase;
                    rr.cnt = mycnt;
                    runs[qat + dcell] = rr;
                }
            }
        }
    };
    if constexpr (W <= 2) {
        // Scalar loads issued by hand, one record ahead: the wait for record i is followed by the load of record
        // i + 1, which then has the whole scoring of record i (~50 VALU instructions) to arrive.  Left to the
        // compiler every iteration was load -> wait -> compute (two dependent round trips: first the stage range, then
        // the operands), and the stage ran at 58 % VALU occupancy with 5-7 waves per SIMD.
        typedef uint32_t qv_t __attribute__((ext_vector_type(8 * W)));
        typedef uint32_t t8_t __attribute__((ext_vector_type(8)));
        typedef uint32_t t4_t __attribute__((ext_vector_type(4)));
        // the 13 tail dwords this kernel reads (RQ_REC_LOWER .. RQ_REC_CELL0) as 8 + 4 + 1 scalar registers instead of a 16-dword
        // tuple: two records in flight then fit 96 scalar registers -- the eighth wave per SIMD (102 were seven)
        struct Tail {
            t8_t a;
            t4_t b;
            uint32_t c;
            __device__ __forceinline__ uint32_t operator[](int i) const { return i < 8 ? a[i] : (i < 12 ? b[i - 8] : c); }
        };
        static_assert(RQ_REC_CELL0 == 12 && RQ_REC_SLOT < 12, "tail fields read through the 8 + 4 + 1 split");
        qv_t qa, qb;
        Tail ta, tb;
#define RQ_SLOAD(Q, T, PTR)                                                                                   \
    do {                                                                                                      \
        if constexpr (W == 1) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(Q) : "s"(PTR));               \
        else asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(Q) : "s"(PTR));                                \
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(T.a) : "s"(PTR), "i"(8 * W * 4));                     \
        asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(T.b) : "s"(PTR), "i"(8 * W * 4 + 32));                \
        asm volatile("s_load_dword %0, %1, %2" : "=s"(T.c) : "s"(PTR), "i"(8 * W * 4 + 48));                  \
    } while (0)
#define RQ_SWAIT(Q, T) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(Q), "+s"(T.a), "+s"(T.b), "+s"(T.c))
        RQ_SLOAD(qa, ta, rec);
        uint32_t i = pb;
        while (true) {
            const uint32_t *nx = i + 1 < pe ? rec + STRIDE : rec;  // the last prefetch re-reads the last record
            RQ_SWAIT(qa, ta);
            RQ_SLOAD(qb, tb, nx);
            score_pair(qa, ta);
            if (++i >= pe) break;
            rec = nx;
            nx = i + 1 < pe ? rec + STRIDE : rec;
            RQ_SWAIT(qb, tb);
            RQ_SLOAD(qa, ta, nx);
            score_pair(qb, tb);
            if (++i >= pe) break;
            rec = nx;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the trailing prefetch has landed before the registers are reused
#undef RQ_SLOAD
#undef RQ_SWAIT
    } else {
        for (uint32_t i = pb; i < pe; ++i, rec += STRIDE) score_pair(rec, rec + 8 * W);
    }
}

