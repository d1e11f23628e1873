// kernels_scan_decl.h -- declarations of the two scan kernel templates (definitions: kernels_scan_valu.h, kernels_scan_mfma.h, each
// compiled in a translation unit of its own -- inst_scan_valu.hip / inst_scan_mfma.hip -- that instantiates every variant the host
// launches; the host side only sees these declarations) and the compile-time geometry the launches need.  Included by
// kernels_query.h after the record layout, ScanArgs and SCAN_PARAMS.
#pragma once

// ---- VALU scan (kernels_scan_valu.h) ----
template <int W, int CPL, bool ARENA = false>
__global__ __launch_bounds__(256) void scan_kernel(SCAN_PARAMS);

// ---- matrix-core scan (kernels_scan_mfma.h) ----
#ifndef RQ_ADD_NT2
#define RQ_ADD_NT2 6  // sub-tiles per wave of the additive-gate instantiation at dim 128: every staged query tile (7 ds_read_b128 per lane)
                      // serves six 32 x 32 steps (four in round 4: launch 6.0 -> 5.8 ms on the headline workload; seven / eight spill inside
                      // the tile loop: 6.6 / 6.8 ms).  Its 33 spilled registers sit in the block's start-up and the cold path only.
#endif
#ifndef RQ_NT_W2
#define RQ_NT_W2 3   // sub-tiles per wave at dim 128, bf16-threshold form (uniform and arena instantiations)
#endif
#ifndef RQ_NT_W12
#define RQ_NT_W12 2  // sub-tiles per wave at dim 768
#endif
#ifndef RQ_A_RING
#define RQ_A_RING 3  // query fragments the slab-outer streamed form requests ahead of their use (1: the compiler's own schedule)
#endif
#ifndef RQ_GATE_DEFER
#define RQ_GATE_DEFER 1  // additive gate of the narrow instantiations: 1 = one branch per query tile (the cold path recomputes the flagged steps), 0 = one per step
#endif
// blocks per CU the register budget is cut for: the resident operands grow with W (6*W*NT dwords for the
// candidates, 6*W for the query tile), so wide vectors run one block per CU with the full 512-register file
template <int W>
constexpr int scan_mfma_blocks_per_cu() { return W <= 2 ? 4 : (W <= 12 ? 2 : 1); }
// Wide vectors (dim >= 384) do not keep the query tile's operand in registers: the candidates' expanded codes
// (6*W*NT dwords) already fill most of the file, so the query fragments are streamed from the LDS image one
// 64-dimension slab at a time (3 ds_read_b64 per slab, each feeding the MFMAs of all NT sub-tiles).  dim 768 runs
// TWO sub-tiles per wave this way (240 VGPRs, two blocks per CU): with one, every 128 candidates re-streamed the list's
// query tiles (20 KB each) from L2 and every MFMA needed its own 1.5 KB fragment from LDS -- both above what the CU's
// L2 and LDS ports deliver at the matrix rate (100M x 768, batch 32 768: 36.2 -> 26.3 ms, 0.21 -> 0.29 of the fp6 peak;
// three sub-tiles at one block per CU: 39 ms).
template <int W>
constexpr bool scan_mfma_stream_a() { return W > 4 && W <= 12; }  // W = 16 (one block per CU) keeps A resident
// Query tiles consumed per block barrier.  The four waves of a block sit on four SIMDs that each serve other
// blocks as well, so a barrier per 32-query tile makes every wave advance at the pace of the slowest; narrow
// vectors (small tile images) afford two tiles per barrier with a 4-slot ring.
template <int W>
constexpr uint32_t scan_mfma_tiles_per_barrier() { return W <= 2 ? 2u : 1u; }
// Waves per block.  The block's waves share every staged query tile, so the L2 -> LDS traffic of a launch is one tile image
// per (candidate tile, query tile): at dim 128 that re-staging was 18 % of the launch (ablation scan_debug bit 1: 8.87 ->
// 7.31 ms; the block barriers, by contrast, cost nothing: bit 13).  Eight waves per block (768 candidates per tile image,
// two blocks per CU: the same 16 waves per CU) halve it.
// (not for the arena instantiations: their exact path and flushes dominate, on the hard distribution eight waves per block
// were 16 % slower; not for dim 768 either, where eight waves mean ONE block per CU and nothing hides a block's start-up:
// 26.0 -> 31.1 ms.)
template <int W, bool ARENA = false>
constexpr int scan_mfma_waves() { return W == 2 && !ARENA ? 8 : 4; }
// Periods of query tiles in flight ahead of the one being consumed (several tiles per barrier only).  Two periods (a
// six-slot ring) were measured for the eight-wave blocks against one on the same box: 8.71 / 8.74 ms against 8.63 / 8.77 --
// the copies are not late, what re-staging costs is their traffic.
template <int W, bool ARENA = false>
constexpr uint32_t scan_mfma_periods_ahead() { return 1u; }
template <int W, bool ARENA = false>
constexpr uint32_t scan_mfma_ring_slots() {
    // one tile per barrier: slots - 1 tiles in flight.  The wide instantiations run one block per CU and wait on the
    // arrival of their (large) tile images, not on the matrix pipe: they take the LDS a second block would have used
    // for a deeper ring
    return scan_mfma_tiles_per_barrier<W>() > 1 ? (1 + scan_mfma_periods_ahead<W, ARENA>()) * scan_mfma_tiles_per_barrier<W>() : (W >= 16 ? 5u : 3u);
}
template <int W, bool ARENA, bool ADD>
constexpr uint32_t scan_mfma_img_dwords() { return rq_img_dwords(12 * W, ADD); }
template <int W, int NT, bool ARENA = false, bool ADD = false>
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
                                                           const ScanArgs a);
