// host_query.h -- part of the host side of librabitq_hip.so (one translation unit: rabitq_hip.hip includes the host_*.h files in order;
// they are not stand-alone headers).  The query pipeline: RaBitQ::query (src/rabitq.rs:268-333) as staged passes -- run_pass, capacities and re-runs, the begin / end halves of a batch call.
#pragma once
// ------------------------------------------------------------------------------------------------
// the query pipeline
// ------------------------------------------------------------------------------------------------
struct QueryParams {
    uint32_t nq, len, probe, topk;
    bool heuristic;
    uint32_t cap, hcap;  // survivor / heuristic-array capacity per query (powers of two)
    // Seeded pass (rq_query_batch_device_seeded): per-query initial thresholds (device; f32::MAX = none).  A first pass
    // runs the whole stream as ONE stage under them; an overflow re-run (row map given) starts from them and stages as usual.
    const float *thr_init = nullptr;
    // Segmented pass: `cap` bounds the stages whose span fits it; a stage that can exceed it appends to the shared arena and
    // its survivors are scattered into per-query segments sized by their exact counts (the workspace then scales with the
    // sum of the survivors instead of nq x the worst query)
    bool seg_final = false;
    bool ext_lists = false;  // the probe lists come from the caller: no coarse ranking in the pass (and no nq x k distance matrix)
};

#define RQ_DEFAULT_CAP 4096u
#define RQ_MAX_CAP_HINT 32768u
#define RQ_MAX_NQ_PER_PASS 65536u
#define RQ_MAX_PROBE 16384u

static rq_status ws_prepare(const rq_index *idx, Workspace &ws, const QueryParams &qp) {
    const uint32_t nprobe = std::min(qp.probe, idx->k);
    const uint64_t nq = qp.nq, npairs = nq * nprobe;
    if (!ws.stream) HIPC(hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking));
    if (!ws.h_totals) HIPC(hipHostMalloc((void **)&ws.h_totals, 16 * sizeof(unsigned long long)));
    RQC(ws.qpad.ensure(nq * idx->dim));
    RQC(ws.y.ensure(nq * idx->dim));
    if (!qp.ext_lists) RQC(ws.dist.ensure(nq * idx->k));
    RQC(ws.probe_dist.ensure(npairs));
    RQC(ws.probe_cluster.ensure(npairs));
    RQC(ws.scal.ensure(npairs));
    if (!scan_is_fused(idx->W)) RQC(ws.planes.ensure(npairs * 4 * idx->W));  // bit planes: only the generic-W scan reads them
    RQC(ws.qnib.ensure(npairs * 8 * idx->W));
    RQC(ws.qf6.ensure(npairs * 12 * idx->W));
    RQC(ws.rough_cnt.ensure(nq));
    RQC(ws.totals.ensure(16));  // [0..7] the pass's totals, [12] rows of the pre-filtered coarse ranking that fell back to exact order
    RQC(ws.stat.ensure(256));
    // record-major (8W + tail per pair) or tile images (pairs padded to 32 per list, 12W + 2 + tail per slot)
    RQC(ws.recs.ensure((npairs + 32ull * idx->k + 32) * (12ull * idx->W + 2 + RQ_REC_TAIL)));
    RQC(ws.grp_cnt.ensure(idx->k + 4));
    RQC(ws.grp_start.ensure(idx->k + 1));
    RQC(ws.q_hist.ensure(idx->k + 2));
    RQC(ws.q_start.ensure(idx->k + 2));
    RQC(ws.q_order.ensure(nq));
    RQC(ws.thr.ensure(nq));
    RQC(ws.surv.ensure(nq * qp.cap));
    RQC(ws.runs.ensure(nq * qp.cap));
    // second run directory, through which long directories (> 512 runs) are ordered: only once the index has shown
    // that it produces them (or with enlarged buffers); until then a stray long directory is bitonic-sorted in place
    ws.use_runs_tmp = qp.cap > RQ_DEFAULT_CAP || qp.seg_final || idx->big_dirs_hint.load() > 0;
    if (ws.use_runs_tmp) RQC(ws.runs_tmp.ensure(nq * qp.cap));
    RQC(ws.surv_cnt.ensure(nq));
    RQC(ws.heap_len.ensure(nq));
    RQC(ws.heap_key.ensure(nq * qp.topk));
    RQC(ws.heap_id.ensure(nq * qp.topk));
    RQC(ws.precise.ensure(nq));
    RQC(ws.need.ensure(nq));
    RQC(ws.ovf.ensure(nq));
    RQC(ws.q_cap.ensure(nq));
    RQC(ws.q_base.ensure(nq));
    RQC(ws.nsurv.ensure(nq));
    RQC(ws.nshadow.ensure(nq));
    RQC(ws.recent.ensure(nq));
    RQC(ws.win_count.ensure(nq));
    RQC(ws.arr_len.ensure(nq));
    RQC(ws.row_map.ensure(nq));
    RQC(ws.big_list.ensure(nq + 3));  // [nq] = entries, [nq + 1] = blocks done, [nq + 2] = most entries of any stage of the pass
    if (qp.heuristic) RQC(ws.arr.ensure(nq * qp.hcap));
    return RQ_OK;
}

struct PassResult {
    uint64_t rough = 0, precise = 0, overflowed = 0, max_need = 0;
};

// Second half of a pass: wait for the stream, read the totals, collect the profile.
static rq_status finish_pass(const rq_index *idx, Workspace &ws, PassResult *res, rq_profile_t *prof_acc) {
    Prof &pf = ws.prof;
    const uint32_t nq = ws.pend_nq, dim = idx->dim;
    HIPC(hipStreamSynchronize(ws.stream));
    HIPC(hipGetLastError());
    res->rough = ws.h_totals[0];
    res->precise = ws.h_totals[1];
    res->overflowed = ws.h_totals[2];
    res->max_need = ws.h_totals[4];
    if (rq_large_batch(nq)) const_cast<rq_index *>(idx)->big_dirs_hint.store((uint32_t)ws.h_totals[7]);
    // The additive gate is a looser test than the rank-5 threshold it replaces: an index / workload on which it sends more than
    // 3 % of the sub-tile steps down the exact path (each costs ~10 plain steps) goes back to the bf16 threshold MFMA for good
    // (results do not depend on the choice; option scan_gate pins it)
    if (ws.pend_additive && ws.h_totals[8] >= 4096 && ws.h_totals[9] * 32 > ws.h_totals[8])
        const_cast<rq_index *>(idx)->additive_loose.store(1);
    if (prof_acc) prof_acc->matrix_subtile_steps += ws.h_totals[8], prof_acc->matrix_exact_steps += ws.h_totals[9];
    if (prof_acc) prof_acc->coarse_fallback_rows += (uint32_t)std::min<unsigned long long>(ws.h_totals[10], 0xFFFFFFFFull);
    if (pf.on && prof_acc) {
        float ms[PF_N] = {0};
        pf.collect(ms);
        prof_acc->ms_rotate += ms[PF_ROTATE], prof_acc->ms_coarse += ms[PF_COARSE];
        prof_acc->ms_select += ms[PF_SELECT], prof_acc->ms_prep += ms[PF_PREP], prof_acc->ms_group += ms[PF_GROUP];
        prof_acc->ms_scan += ms[PF_SCAN] + ms[PF_SCAN_MATRIX], prof_acc->ms_scan_matrix += ms[PF_SCAN_MATRIX];
        prof_acc->ms_rerank += ms[PF_RERANK], prof_acc->ms_sort += ms[PF_SORT];
        if (!ws.pend_matrix_ranges.empty()) {  // pairs scored by those launches: per query, its stream length clipped to the range
            std::vector<unsigned long long> len(nq);
            HIPC(hipMemcpy(len.data(), ws.rough_cnt.p, (size_t)nq * 8, hipMemcpyDeviceToHost));
            for (const StreamRange &r : ws.pend_matrix_ranges)
                for (uint32_t b = 0; b < nq; ++b)
                    prof_acc->matrix_pairs += std::min<unsigned long long>(len[b], r.s_hi) - std::min<unsigned long long>(len[b], r.s_lo);
        }
        prof_acc->ms_replay += ms[PF_REPLAY], prof_acc->ms_total += ms[PF_TOTAL], prof_acc->ms_early += ms[PF_EARLY];
    }
    if (prof_acc) {
        const uint64_t slots = std::max<uint64_t>((uint64_t)nq * ws.pend_cap, ws.pend_seg_slots);
        prof_acc->survivor_workspace_bytes = std::max<uint64_t>(prof_acc->survivor_workspace_bytes,
            ws.pend_seg_slots ? (ws.surv.count + ws.runs.count + ws.runs_tmp.count + ws.arena_recs.count + ws.arena_runs.count) * 16ull
                              : slots * (ws.use_runs_tmp ? 48ull : 32ull));
        prof_acc->segmented_passes += ws.pend_seg_slots ? 1u : 0u;
        prof_acc->scan_candidates += res->rough;
        prof_acc->scan_bytes += res->rough * (uint64_t)(dim / 8 + 16);
        prof_acc->rerank_candidates += ws.h_totals[3];
        prof_acc->rerank_shadow_rejects += ws.h_totals[5];
        if (g_scan_dbg.load() & 256) {  // developer hook: where the matrix-core scan's waves spend their cycles
            unsigned long long ht[12];
            HIPC(hipMemcpy(ht, ws.stat.p + 128, sizeof ht, hipMemcpyDeviceToHost));
            if (ht[3])
                fprintf(stderr, "[rabitq_hip] scan_mfma exact path (wave 0 of every block): %.0f cycles per block inside it (of which flushes %.0f), "
                        "%.1f flagged registers, %.1f with survivors and %.2f flushes per block\n", (double)ht[5] / ht[3], (double)ht[6] / ht[3],
                        (double)ht[9] / ht[3], (double)ht[7] / ht[3], (double)ht[8] / ht[3]);
            if (ht[3])
                fprintf(stderr, "[rabitq_hip] scan_mfma timing: %llu blocks, %.1f tiles/block; per block cycles: start-up %.0f, "
                        "tile-loop waits %.0f, tile bodies %.0f (per tile: wait %.0f, body %.0f)\n", ht[3], (double)ht[4] / ht[3],
                        (double)ht[0] / ht[3], (double)ht[1] / ht[3], (double)ht[2] / ht[3], (double)ht[1] / std::max(1ull, ht[4]),
                        (double)ht[2] / std::max(1ull, ht[4]));
        }
        if ((g_scan_dbg.load() & 4096) && nq <= RQ_SB_MAX_NQ) {  // developer hook: phase boundaries of sb_query_kernel's block 0
            unsigned long long hs[32];
            HIPC(hipMemcpy(hs, ws.stat.p, sizeof hs, hipMemcpyDeviceToHost));
            std::string line = "[rabitq_hip] sb_query_kernel phases (us since entry):";
            for (unsigned long long i = 1; i < std::min<unsigned long long>(hs[0], 30); ++i) line += " " + std::to_string((hs[1 + i] - hs[1]) / 100.0).substr(0, 6);
            fprintf(stderr, "%s\n", line.c_str());
        }
    }
    return RQ_OK;
}

// Runs one pass over nq queries already resident at d_q (nq x len).  Results go to row
// row_map[b] (or b) of the output arrays.  On return the stream is synchronised.
// ext_cluster / ext_dist (nq x min(probe,k), device): if given, the probe lists are taken from there
// (visiting order as supplied; id 0xFFFFFFFF = no list) instead of being ranked here.
static rq_status run_pass(const rq_index *idx, Workspace &ws, const float *d_q, const QueryParams &qp,
                          const uint32_t *d_row_map, float *d_out_dist, uint32_t *d_out_id, uint32_t *d_out_n,
                          PassResult *res, rq_profile_t *prof_acc, const uint32_t *ext_cluster = nullptr,
                          const float *ext_dist = nullptr, bool defer = false) {
    const uint32_t dim = idx->dim, k = idx->k, W = idx->W;
    const uint32_t nq = qp.nq, nprobe = std::min(qp.probe, k), topk = qp.topk;
    const uint32_t npairs = nq * nprobe;
    bool listed = false;  // sharded pass: the pairs whose list has members here are listed (ws.live_list, nlive of them)
    uint32_t nlive = 0;
    hipStream_t st = ws.stream;
    ws.pend_prefiltered = false;
    Prof &pf = ws.prof;
    pf.reset(g_profiling.load(), st);
    pf.begin(PF_TOTAL);
    size_t total_span = pf.spans.size() ? pf.spans.size() - 1 : 0;

    const int impl = g_scan_impl.load();  // one consistent choice for the whole pass
    // Stream stages.  The reference visits a query's candidates as ONE stream: probed lists nearest-first,
    // members in stored order.  A stage covers stream positions [s_lo, s_hi) (of every query) and is
    // scanned with the threshold each query's ranker holds at the start of the stage -- an upper
    // bound of the reference's threshold everywhere in the stage, since it never rises -- then the
    // survivors are replayed in the reference's order.  Stage 0 = the first topk candidates
    // (threshold f32::MAX), later stages grow geometrically.
    struct Stage {
        uint32_t s_lo, s_hi;
    };
    // first_hi: end of the first stage; settle_cap: where the early stages must end at the latest
    auto build_stages = [&](uint64_t first_hi, uint64_t growth, uint64_t settle_cap) {
        std::vector<Stage> stages;
        const uint64_t total_max = std::min<uint64_t>((uint64_t)nprobe * idx->max_list_len, idx->n);
        uint64_t lo = 0, hi = first_hi;
        const uint64_t avg = std::max<uint64_t>(1, idx->n / std::max<uint32_t>(idx->nonempty_lists, 1));  // (over the lists that exist here: a shard owns k / world of them)
        // the threshold has settled once a query has seen its whole nearest list; with unbalanced lists (Zipf sizes) the
        // nearest list of many queries is one of the long ones, so the bar is the LONGEST list (capped: a single
        // monster list must not push the whole batch through many thin stages)
        uint64_t settle = std::min(settle_cap, std::max<uint64_t>(avg, std::min<uint64_t>(idx->max_list_len, 16 * avg)));
        if (rq_large_batch(nq)) settle = std::max<uint64_t>(1, settle * (uint64_t)g_stage_settle_pct.load() / 100);
        while (lo < total_max) {
            // past the first two lists' worth of candidates the threshold is already tight: scan the rest of
            // the stream as ONE stage (every list then meets all its queries at once: full 32-query tiles)
            const bool last = hi >= total_max || lo >= settle;
            stages.push_back({(uint32_t)lo, last ? 0xFFFFFFFFu : (uint32_t)hi});
            if (last) break;
            lo = hi;
            hi = std::min<uint64_t>(hi * growth, 0xFFFFFFF0ull);
            // the geometric step must not carry an early (VALU) stage over many lists when lists are short:
            // past two lists' worth the rest belongs to the final stage
            // (a step that ends within a factor two BELOW that mark is carried up to it: the hard distribution ran a thin matrix-core
            // stage [40960, 48828) behind [5120, 40960) -- 4.5 ms of launches for 8 000 stream positions)
            if (lo < 2 * avg && 2 * hi > 2 * avg) hi = 2 * avg;
            // ... and the last early stage ends exactly where the threshold has settled: everything beyond belongs to the
            // final (matrix-core) stage, where a list meets all its queries at once
            if (lo < settle && hi > settle) hi = settle;
        }
        return stages;
    };
    std::vector<Stage> stages;
    ReplayState rs;
    rs.thr = ws.thr.p, rs.heap_len = ws.heap_len.p, rs.heap_key = ws.heap_key.p, rs.heap_id = ws.heap_id.p;
    rs.precise = ws.precise.p, rs.need = ws.need.p, rs.nsurv = ws.nsurv.p, rs.nshadow = ws.nshadow.p, rs.recent_max = ws.recent.p, rs.win_count = ws.win_count.p;
    rs.arr_len = ws.arr_len.p, rs.arr = ws.arr.p, rs.hcap = qp.hcap;
    rs.ovf = ws.ovf.p;
    const QSeg useg{nullptr, nullptr, qp.cap};  // uniform geometry: every stage but a segmented final one
    const float *qpad = d_q;
    const uint32_t *probe_cluster = ws.probe_cluster.p;
    const float *probe_dist = ws.probe_dist.p;
    const uint32_t *rerank_order = nullptr;
    const bool one_stage = qp.thr_init != nullptr && d_row_map == nullptr;  // thresholds are already tight: nothing to learn in early stages
    // matrix cores pay once many queries share each list AND survivors are rare, i.e. past the nearest list
    // (stages inside it leave hundreds of survivors per query: the exact path dominates there and the VALU
    // kernel wins, measured at any batch size)
    auto stage_on_matrix = [&](const Stage &sg) {
        const uint64_t avg_len = std::max<uint64_t>(1, idx->n / std::max<uint32_t>(idx->nonempty_lists, 1));
        const uint64_t span = (uint64_t)std::min<uint64_t>(sg.s_hi, (uint64_t)nprobe * idx->max_list_len) - sg.s_lo;
        const uint64_t est_pairs = (uint64_t)nq * std::min<uint64_t>(nprobe, span / avg_len + 2);
        return scan_has_mfma(W) && impl != 1 &&
               (impl == 2 || (est_pairs >= 8ull * k && (sg.s_lo >= avg_len * (uint64_t)g_stage_settle_pct.load() / 100 || one_stage)));
    };

    // ---- small batches: few, fat launches (kernels_small.h) -------------------------------------------------------
    const bool sb_w = W == 1 || W == 2 || W == 4 || W == 8 || W == 12 || W == 16;
    bool small = g_small_batch.load() == 0 && nq <= RQ_SB_MAX_NQ && !ext_cluster && !d_row_map && !qp.thr_init && sb_w &&
                 k <= RQ_SB_MAX_K && nprobe <= 64 && topk <= RQ_SB_MAX_TOPK && qp.cap <= 4 * RQ_DEFAULT_CAP;
    bool sb_results_done = false;   // results and totals were written by the small-batch kernels (heap ranker)
    bool sb_fused_finish = false;   // the final stage ends in sb_finish_kernel
    bool sb_filled = false;         // the final stage's pair-major records were written by sb_query_kernel
    if (small) {
        // the early stages run inside one block per query: the first one takes what would be two (16 x topk candidates
        // under threshold f32::MAX cost one gather round), and the in-block part ends after 64 K candidates at the latest
        const int gopt = g_stage_growth.load();
        stages = build_stages(16ull * std::max<uint32_t>(topk, 1), gopt >= 2 ? (uint64_t)gopt : 8, (uint64_t)std::max(1, g_sb_span.load()));
        if (stages.size() > RQ_SB_MAX_STAGES) small = false;
    }
    if (small) {
        const uint64_t total_max = std::min<uint64_t>((uint64_t)nprobe * idx->max_list_len, idx->n);
        SbArgs sa{};
        // a short remainder (small indexes, few probes) is scanned in the block as well: no further launch
        const bool whole = stages.empty() || (total_max - stages.back().s_lo) * (uint64_t)(dim / 8 + 16) <= (1ull << 20);
        sa.nstages = (uint32_t)(whole ? stages.size() : stages.size() - 1);
        for (uint32_t i = 0; i < sa.nstages; ++i) sa.s_lo[i] = stages[i].s_lo, sa.s_hi[i] = stages[i].s_hi;
        const Stage fin = whole ? Stage{0, 0} : stages.back();
        const uint64_t fin_pairs = (uint64_t)nq * nprobe;
        const bool fin_cluster_major = !whole && fin_pairs >= k / 2 && fin_pairs > 64;  // the stage loop's own rule for a full-probe stage
        sa.finalize = whole ? 1u : 0u;
        sa.fill_final = !whole && !fin_cluster_major ? 1u : 0u;
        sa.final_lo = fin.s_lo;
        sa.codes = reinterpret_cast<const uint32_t *>(idx->codes.p), sa.factors = idx->factors.p, sa.centroids = idx->centroids.p;
        sa.offsets = idx->offsets.p, sa.map_ids = idx->map_ids.p, sa.base = idx->view();
        sa.dist = ws.dist.p, sa.y = ws.y.p, sa.qpad = ws.qpad.p, sa.probe_cluster = ws.probe_cluster.p, sa.probe_dist = ws.probe_dist.p;
        sa.qf6 = scan_has_mfma(W) && impl != 1 ? ws.qf6.p : nullptr;  // a final stage over few lists may run on the matrix cores
        sa.scal = ws.scal.p, sa.qnib = ws.qnib.p, sa.rough_cnt = ws.rough_cnt.p, sa.surv_cnt = ws.surv_cnt.p, sa.totals = ws.totals.p;
        sa.rs = rs, sa.out_dist = d_out_dist, sa.out_id = d_out_id, sa.out_n = d_out_n, sa.recs = ws.recs.p, sa.fs = idx->fstats;
        sa.k = k, sa.dim = dim, sa.nprobe = nprobe, sa.topk = topk, sa.cap = qp.cap, sa.hcap = qp.hcap;
        sa.stamps = (g_scan_dbg.load() & 4096) ? ws.stat.p : nullptr;
        if (sa.stamps) HIPC(hipMemsetAsync(ws.stat.p, 0, 8, st));
        else HIPC(hipMemsetAsync(ws.stat.p, 0, 256 * sizeof(unsigned long long), st));  // (the counters of a final matrix-core stage, if any)
        pf.begin(PF_COARSE);
        sb_front_kernel<<<dim3(ceil_div(k, RQ_SB_LISTS), ceil_div(nq, RQ_SB_QT)), 256, (size_t)2 * RQ_SB_QT * dim * sizeof(float), st>>>(
            d_q, qp.len, idx->P.p, idx->centroids.p, ws.y.p, ws.qpad.p, ws.dist.p, k, dim, nq, ws.totals.p, ws.big_list.p + nq);
        pf.end();
        pf.begin(PF_EARLY);
        const size_t dyn = (size_t)RQ_SB_CAP * sizeof(SurvRec) + (size_t)dim * 4 + (size_t)topk * 16;
        const int mode = qp.heuristic ? 2 : (topk < 64 ? 1 : 0);
#define RQ_SBQ(WW)                                                                        \
    do {                                                                                  \
        if (mode == 2) sb_query_kernel<WW, 2><<<nq, 1024, dyn, st>>>(sa);                 \
        else if (mode == 1) sb_query_kernel<WW, 1><<<nq, 1024, dyn, st>>>(sa);            \
        else sb_query_kernel<WW, 0><<<nq, 1024, dyn, st>>>(sa);                           \
    } while (0)
        switch (W) {
            case 1: RQ_SBQ(1); break;
            case 2: RQ_SBQ(2); break;
            case 4: RQ_SBQ(4); break;
            case 8: RQ_SBQ(8); break;
            case 12: RQ_SBQ(12); break;
            default: RQ_SBQ(16); break;
        }
#undef RQ_SBQ
        pf.end();
        qpad = ws.qpad.p;
        sb_results_done = whole && !qp.heuristic;
        sb_fused_finish = !whole && !qp.heuristic;
        sb_filled = sa.fill_final != 0;
        stages.clear();
        if (!whole) stages.push_back(fin);
        if (prof_acc) prof_acc->small_batch_passes++;
    } else {
    // 1. pad (rabitq.rs:277-280) + rotate (:282)
    pf.begin(PF_ROTATE);
    if (qp.len != dim) {
        pad_rows_kernel<<<ceil_div((uint64_t)nq * dim, 256), 256, 0, st>>>(d_q, ws.qpad.p, nq, qp.len, dim);
        qpad = ws.qpad.p;
    }
    launch_rotate(qpad, idx->P.p, ws.y.p, nq, dim, nq >= 32, st);
    pf.end();

    // 2. coarse distances + probe selection (:283-297)
    if (ext_cluster) {
        probe_cluster = ext_cluster;
        probe_dist = ext_dist;
    } else {
        if (coarse_prefilter_applies(idx, nq, nprobe)) {
            pf.begin(PF_COARSE);
            HIPC(hipMemsetAsync(ws.totals.p + 12, 0, 8, st));
            RQC(ws.coarse_redo.ensure(nq));
            RQC(ws.qf6.ensure((size_t)nq * dim / 2 + 16));  // (room for the pre-rounded query rows of the wide instantiation)
            launch_coarse_prefiltered(idx, ws.y.p, ws.dist.p, nq, nprobe, ws.probe_cluster.p, ws.probe_dist.p, nprobe, ws.totals.p + 12, ws.coarse_redo.p, st,
                                      reinterpret_cast<uint16_t *>(ws.qf6.p));  // (the fp6 images are written later: prep)
            ws.pend_prefiltered = true;
            pf.end();
        } else {
            pf.begin(PF_COARSE);
            launch_coarse(idx->cent_t.p, ws.y.p, ws.dist.p, k, dim, nq, k, st);
            pf.end();
            pf.begin(PF_SELECT);
            launch_select(ws.dist.p, k, nprobe, ws.probe_cluster.p, ws.probe_dist.p, 0, nprobe, nq, st);
            pf.end();
        }
    }

    // the pass's stages, planned before the quantisation (which writes the VALU scans' operand only for the probe slots an early
    // stage can reach)
    if (one_stage) {
        stages.push_back({0u, 0xFFFFFFFFu});
    } else {
        // geometric growth of the early stages: 16 (coarser stages: a stage of launches less) below 32 768 queries, 8 (tighter
        // thresholds: ~9 % fewer exact distances) from there on.  Up to round 4 the step to 8 came at 256 queries; re-swept on the
        // round-5 kernels: 512 queries 1.70 -> 1.54 ms per call with 16, 2048 2.32 -> 2.23, 8192 3.83 -> 3.74, 16 384 5.64 -> 5.57,
        // 65 536 17.05 -> 17.12
        const int gopt = g_stage_growth.load();
        const uint64_t growth = gopt >= 2 ? (uint64_t)gopt : (nq >= 32768 ? 8 : 16);
        // the first stage runs with threshold f32::MAX (everything survives) until the ranker's heap is full; in a large
        // batch it also takes what would be the next stage (whose threshold -- the worst of the first topk -- lets most
        // of it through anyway): one stage of launches less for ~1 % more exact distances
        stages = build_stages((uint64_t)std::max<uint32_t>(topk, 1) * (rq_large_batch(nq) ? growth : 1), growth, ~0ull);
    }
    // 3. per-pair query quantisation (:304-317)
    pf.begin(PF_PREP);
    {
        uint32_t *qn = scan_is_fused(W) ? ws.qnib.p : nullptr;
        uint32_t *q6 = scan_has_mfma(W) && impl != 1 ? ws.qf6.p : nullptr;
        // The 4-bit operand (64 of a pair's ~210 bytes at dim 128) is read by the VALU scans only, and a VALU stage that ends at stream
        // position s_hi cannot reach probe slot s_hi / (shortest list) or beyond: the matrix-core stages' pairs are written without it.
        uint32_t qn_slots = nprobe;
        if (qn && q6 && !ext_cluster && idx->min_list_len > 0) {
            uint32_t reach = 0;
            for (const Stage &sg : stages)
                if (!stage_on_matrix(sg))
                    reach = std::max<uint32_t>(reach, sg.s_hi == 0xFFFFFFFFu ? nprobe : (uint32_t)std::min<uint64_t>(nprobe, (uint64_t)(sg.s_hi - 1) / idx->min_list_len + 1));
            qn_slots = reach;
        }
        // an index most of whose lists are empty (a shard of a multi-GPU deployment: the probe lists name the lists of every shard):
        // the pairs with nothing to scan are settled by one thread each, the quantisation runs over the listed others
        listed = idx->nonempty_lists * 2 < k && npairs >= 65536 && g_pair_split.load() != 0 &&
                 (dim == 64 || dim == 128 || dim == 256 || dim == 512 || dim == 768 || dim == 1024);
        if (listed) {
            RQC(ws.live_list.ensure((size_t)npairs + 1));
            HIPC(hipMemsetAsync(ws.live_list.p + npairs, 0, 4, st));
            pair_split_kernel<<<ceil_div(npairs, 4096), 1024, 0, st>>>(idx->offsets.p, probe_cluster, probe_dist, npairs, nprobe, k, ws.scal.p,
                                                                       ws.live_list.p, ws.live_list.p + npairs);
            // the launches over the listed pairs are sized by their number: one small copy and a wait (tens of microseconds against
            // the milliseconds that 7 of 8 idle lane groups cost)
            // (into the workspace's pinned block, not a pageable stack word; the wait is the price of sizing the launches below by the
            // count -- rq_query_batch_device_begin on a shard-like index therefore returns only once rotate, coarse ranking and this
            // split have run: include/rabitq_hip.h says so)
            unsigned long long *h_live = ws.h_totals + 15;
            *h_live = 0;
            HIPC(hipMemcpyAsync(h_live, ws.live_list.p + npairs, 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            nlive = (uint32_t)*h_live;
        }
#define RQ_PREP_SMALL(LP, R, PPB, PP)                                                                              \
    do {                                                                                                           \
        if (listed)                                                                                                \
            prep_small_listed_kernel<LP, R, PP><<<std::max(1u, ceil_div(nlive, (PPB) * (PP))), 256, 0, st>>>(ws.y.p, idx->centroids.p, idx->offsets.p, probe_cluster, \
                                                                    probe_dist, ws.live_list.p, nlive, nprobe, ws.scal.p, qn, q6, k); \
        else                                                                                                       \
            prep_small_kernel<LP, R, PP><<<ceil_div(npairs, (PPB) * (PP)), 256, 0, st>>>(ws.y.p, idx->centroids.p, idx->offsets.p, probe_cluster, \
                                                                    probe_dist, npairs, nprobe, ws.scal.p, qn, q6, k, idx->nonempty_lists * 2 < k ? 2u : 1u, qn_slots); \
    } while (0)
        // dim 128: 16 lanes per pair, two rounds of 64 dimensions, two pairs per lane group in flight (round 4: 32 lanes, one round, four
        // pairs: the min / max / sum reductions over the pair's lanes are half of the kernel's vector work, and half the lanes do a
        // quarter less of it: 0.96 -> 0.66 ms per 4.2 M pairs)
        if (dim == 128) RQ_PREP_SMALL(16, 2, 16, 2);
        else if (dim == 64) RQ_PREP_SMALL(16, 1, 16, 4);
        else if (dim == 256) RQ_PREP_SMALL(32, 2, 8, 2);
        else if (dim == 512) RQ_PREP_SMALL(32, 4, 8, 1);
        else if (dim == 768) RQ_PREP_SMALL(32, 6, 8, 1);  // (1.61 -> 1.34 ms per 2.1 M pairs against 64 lanes x 3 rounds x 2 pairs)
        else if (dim == 1024) RQ_PREP_SMALL(32, 8, 8, 1);
#undef RQ_PREP_SMALL
        else
            prep_kernel<<<ceil_div(npairs, 4), 256, 0, st>>>(ws.y.p, idx->centroids.p, idx->offsets.p, probe_cluster, probe_dist,
                                                             npairs, nprobe, dim, ws.scal.p,
                                                             scan_is_fused(W) ? nullptr : ws.planes.p,   // only the generic-W scan reads bit planes
                                                             qn, q6, nullptr, k, 1u);
    }
    pair_prefix_kernel<<<ceil_div(nq, 4), 256, 0, st>>>(ws.scal.p, nq, nprobe, ws.rough_cnt.p);
    if (rq_large_batch(nq)) {  // large batch: rerank queries of the same nearest list back to back (cache locality of the row gather)
        HIPC(hipMemsetAsync(ws.q_hist.p, 0, (size_t)(k + 2) * 4, st));
        order_count_kernel<<<ceil_div(nq, 256), 256, 0, st>>>(probe_cluster, nprobe, nq, k, ws.q_hist.p);
        group_scan_kernel<<<1, 1024, 0, st>>>(ws.q_hist.p, k + 1, ws.q_start.p, 0u, nullptr, 0u);  // also zeroes the histogram: cursor
        order_scatter_kernel<<<ceil_div(nq, 256), 256, 0, st>>>(probe_cluster, nprobe, nq, k, ws.q_start.p, ws.q_hist.p,
                                                                ws.q_order.p);
        rerank_order = ws.q_order.p;
    }
    // 4. ranker state (rerank.rs:70-77, :129-139) and per-query counters
    init_state_kernel<<<ceil_div(nq, 256), 256, 0, st>>>(rs, ws.surv_cnt.p, nq, qp.thr_init, d_row_map);
    HIPC(hipMemsetAsync(ws.totals.p, 0, 8 * sizeof(unsigned long long), st));
    HIPC(hipMemsetAsync(ws.big_list.p + nq, 0, 12, st));
    HIPC(hipMemsetAsync(ws.stat.p, 0, 256 * sizeof(unsigned long long), st));  // the matrix-core scan's step counters (+ developer hooks)
    pf.end();

    // 5. stages: planned above
    }  // !small
    ws.pend_matrix_ranges.clear();
    ws.pend_seg_slots = 0;
    ws.pend_additive = false, ws.pend_matrix_stages = 0;
    // persistent blocks of the long-directory ordering: sized by how many such directories recent passes produced
    const uint32_t big_hint = idx->big_dirs_hint.load();
    const uint32_t mid_blocks = big_hint == 0 ? 64u : std::min(4096u, std::max(256u, big_hint / 4));
    const uint32_t tile = scan_tile(W);
    const uint64_t avg_len = std::max<uint64_t>(1, idx->n / std::max<uint32_t>(idx->nonempty_lists, 1));
    uint32_t stage_no = ~0u;
    for (const Stage &sg : stages) {
        ++stage_no;
        const uint64_t span = (uint64_t)std::min<uint64_t>(sg.s_hi, (uint64_t)nprobe * idx->max_list_len) - sg.s_lo;
        const uint64_t est_pairs = (uint64_t)nq * std::min<uint64_t>(nprobe, span / avg_len + 2);
        // matrix cores pay once many queries share each list AND survivors are rare, i.e. past the nearest list
        // (stages inside it leave hundreds of survivors per query: the exact path dominates there and the VALU
        // kernel wins, measured at any batch size)
        const bool use_mfma = stage_on_matrix(sg);
        // list-major once the stage's pairs reach k / 32 (k / 2 up to round 4, and still on the small-batch path, whose kernels decide with
        // that rule): a pair-major EARLY stage launches a block for every (query, probe slot, tile) although only the first slots are in
        // it -- at 512 queries the early stages took 1.06 ms pair-major against 0.3 list-major (batch 256: 1.43 -> 1.12 ms per call,
        // 512: 2.42 -> 1.62)
        const bool cluster_major = use_mfma || (est_pairs >= k / (small ? 2u : (uint32_t)g_cluster_major_div.load()) && est_pairs > 64);
        if (g_scan_dbg.load() & 16384)  // developer hook: the pass's stage list
            fprintf(stderr, "[rabitq_hip] stage %u: [%u, %u) span %llu est_pairs %llu %s\n", stage_no, sg.s_lo, sg.s_hi,
                    (unsigned long long)span, (unsigned long long)est_pairs, use_mfma ? "matrix cores" : (cluster_major ? "VALU, list-major" : "VALU, pair-major"));
        const bool fp6_records = use_mfma;
        // (an arena stage, below: a stage that can exceed the uniform survivor capacity; its scan instantiation has its own tile)
        const bool arena_stage = qp.seg_final && span > qp.cap && scan_is_fused(W) && rq_large_batch(nq);
        const int gate_opt = g_scan_gate.load();
        const bool additive = use_mfma && !arena_stage && scan_has_additive(W) && idx->list_uref.p != nullptr && gate_opt != 1 &&
                              (gate_opt == 2 || !idx->additive_loose.load());
        pf.begin(PF_GROUP);
        ScanArgs a{};
        ScanPtrs sp{};
        a.cluster_major = cluster_major ? 1u : 0u;
        // slots a stage can touch: slot s starts at stream position >= s * (shortest list), so only the first few
        // slots of every query need to be looked at in the early stages (not derivable when lists may be empty,
        // e.g. a shard that does not own every probed list)
        uint32_t slot_hi = nprobe;
        if (cluster_major && idx->min_list_len > 0 && !ext_cluster && sg.s_hi != 0xFFFFFFFFu)
            slot_hi = (uint32_t)std::min<uint64_t>(nprobe, (uint64_t)(sg.s_hi - 1) / idx->min_list_len + 1);
        const uint32_t stage_pairs = nq * slot_hi;
        bool ranked = false;
        if (cluster_major) {
            HIPC(hipMemsetAsync(ws.grp_cnt.p, 0, (size_t)((k + 4) & ~3u) * 4, st));  // 16-byte multiple: one fill kernel
            // big stages: places inside the groups come out of the counting pass (LDS histogram per block)
            const int rank_opt = g_group_rank.load();  // 0 never, 1 auto, 2 whenever the histogram fits LDS (tests)
            ranked = k <= 32768 && (rank_opt == 2 || (rank_opt == 1 && stage_pairs >= 16 * RQ_RANK_ITEMS &&
                                                      stage_pairs / RQ_RANK_ITEMS >= k / 256));
            if (ranked) {
                const uint32_t nblk = ceil_div(stage_pairs, RQ_RANK_ITEMS);
                RQC(ws.pair_rank.ensure(stage_pairs));
                RQC(ws.rank_base.ensure((size_t)nblk * k));
                group_rank_kernel<<<nblk, 1024, (size_t)k * 4, st>>>(ws.scal.p, probe_cluster, stage_pairs, nprobe, slot_hi, sg.s_lo,
                                                                    sg.s_hi, k, ws.grp_cnt.p, ws.pair_rank.p, ws.rank_base.p);
            } else
                group_count_kernel<<<ceil_div(stage_pairs, 256), 256, 0, st>>>(ws.scal.p, probe_cluster, stage_pairs, nprobe,
                                                                               slot_hi, sg.s_lo, sg.s_hi, ws.grp_cnt.p);
            group_scan_kernel<<<1, 1024, 0, st>>>(ws.grp_cnt.p, k, ws.grp_start.p, (use_mfma ? 1u : 0u) | (ranked ? 2u : 0u) | (additive ? 4u : 0u), ws.recs.p, 12 * W);
            a.ngroups = k;
        } else {
            a.ngroups = npairs;
        }
        // pack the stage's work records (query operand + scalars + current threshold + local range)
        const uint32_t *operand = fp6_records ? ws.qf6.p
                                              : (scan_is_fused(W) ? ws.qnib.p : reinterpret_cast<const uint32_t *>(ws.planes.p));
        if (!(sb_filled && !cluster_major)) {  // (the small-batch kernel has written a pair-major final stage's records already)
            // a sharded pass visits only the listed pairs when the stage's work items ARE the pairs (every slot can be in the stage)
            const bool fill_listed = listed && ranked && cluster_major && slot_hi == nprobe;
            const uint32_t fill_items = fill_listed ? nlive : stage_pairs;
            // eight lanes per pair (16-byte copies) wherever the operand rows are 16-byte aligned: record-major records and the additive
            // tile images (the bf16-form images keep rows of opdw + 2 dwords: 8-byte aligned, sixteen lanes); final stage of the
            // headline step 0.75 -> 0.45 ms
            const bool fill8 = !use_mfma || additive;
            if (fill_items) {
                if (fill8)
                    stage_fill_kernel<8><<<ceil_div(fill_items, 32), 256, 0, st>>>(ws.scal.p, probe_cluster, operand, ws.thr.p, fill_items,
                                                                    nprobe, slot_hi, fp6_records ? 12 * W : 8 * W, sg.s_lo, sg.s_hi,
                                                                    a.cluster_major, ws.grp_start.p, ws.grp_cnt.p, ws.recs.p,
                                                                    idx->fstats, use_mfma ? (additive ? 2u : 1u) : 0u, ranked ? ws.pair_rank.p : nullptr,
                                                                    ws.rank_base.p, k, idx->list_uref.p, fill_listed ? ws.live_list.p : nullptr);
                else
                    stage_fill_kernel<16><<<ceil_div(fill_items, 16), 256, 0, st>>>(ws.scal.p, probe_cluster, operand, ws.thr.p, fill_items,
                                                                    nprobe, slot_hi, fp6_records ? 12 * W : 8 * W, sg.s_lo, sg.s_hi,
                                                                    a.cluster_major, ws.grp_start.p, ws.grp_cnt.p, ws.recs.p,
                                                                    idx->fstats, use_mfma ? (additive ? 2u : 1u) : 0u, ranked ? ws.pair_rank.p : nullptr,
                                                                    ws.rank_base.p, k, idx->list_uref.p, fill_listed ? ws.live_list.p : nullptr);
            }
        }
        if (additive) {  // the stage's v' ranges per list (the candidates' side of the additive bound is built from them in the scan)
            RQC(ws.grp_vref.ensure(2 * (size_t)k));
            group_vrange_kernel<<<k, 256, 0, st>>>(ws.recs.p, ws.grp_start.p, ws.grp_cnt.p, 12 * W, ws.grp_vref.p);
            ws.pend_additive = true;
        }
        pf.end();
        sp.codes = reinterpret_cast<const uint32_t *>(idx->codes.p);
        sp.factors = idx->factors.p;
        sp.grp_start = ws.grp_start.p;
        sp.grp_cnt = ws.grp_cnt.p;
        sp.offsets = idx->offsets.p;
        sp.recs = ws.recs.p;
        sp.surv = ws.surv.p;
        sp.runs = ws.runs.p;
        sp.surv_cnt = ws.surv_cnt.p;
        sp.stat = ws.stat.p;  // 64 x {sub-tile steps, exact-path steps} of the matrix-core scan
        sp.list_uref = idx->list_uref.p, sp.grp_vref = ws.grp_vref.p;
        a.cap = qp.cap;
        a.dbg = (uint32_t)g_scan_dbg.load();
        const uint32_t stage_tile = use_mfma ? scan_mfma_tile(W, arena_stage, additive) : tile;
        a.tiles_per_group = ceil_div(std::min<uint64_t>(idx->max_list_len, sg.s_hi), stage_tile);
        sp.tile_table = nullptr;
        const uint64_t grid_blocks = (uint64_t)k * a.tiles_per_group, real_tiles = idx->n / stage_tile + k;
        const int tt_opt = g_scan_tile_table.load();  // 0 = never, 1 = when the plain grid is mostly empty blocks, 2 = always
        if (cluster_major && scan_is_fused(W) && sg.s_hi >= idx->max_list_len &&
            (tt_opt == 2 || (tt_opt == 1 && grid_blocks > 4 * real_tiles))) {
            // the stage reaches every position of the lists and the lists are very unequal (one block per existing
            // (list, tile) instead of k x the longest list's tiles; measured neutral-to-slower for moderately unequal
            // lists, where the empty blocks of the plain grid cost less than the table's dependent load)
            uint32_t count = 0;
            sp.tile_table = get_tile_table(idx, stage_tile, &count);
            if (sp.tile_table) a.use_table = 1u, a.ngroups = count, a.tiles_per_group = 1u;
        }
        // large batches, VALU-kernel stages: the run descriptors go into a dense directory indexed by stream position
        // (stage_fill_kernel: RQ_REC_CELL0), so the stage needs no sort of its run directory
        uint32_t dense_cells = 0;
        if (rq_large_batch(nq) && !use_mfma && scan_is_fused(W) && g_dense_dir.load() && sg.s_hi != 0xFFFFFFFFu &&
            !(qp.seg_final && span > qp.cap)) {  // (an arena stage appends its runs: they are placed by the scatter pass)
            const uint64_t cells = (uint64_t)((sg.s_hi - 1) >> 6) - (sg.s_lo >> 6) + 2ull * slot_hi + 2;
            if (cells <= qp.cap) dense_cells = (uint32_t)cells;
        }
        a.dense_dir = dense_cells ? 1u : 0u;
        // Arena stage (large batches of an index whose survivor counts are very unequal -- hard distribution, final stage:
        // median 12 survivors per query, mean 2 800, maximum beyond 100 000): every stage that CAN exceed the uniform capacity
        // (span > capacity) appends its survivors to one arena shared by all queries while counting them per query; the
        // exact counts size a segment per query (prefix sum), the host makes room for their sum, and a scatter pass moves
        // every run to its query's segment.  The workspace follows the SUM of the survivors, not nq x the worst query, and
        // no query can overflow.
        QSeg seg = useg;
        bool runs_in_tmp = false;
        if (arena_stage) {
            // capacity: what earlier batches needed (+ headroom), at least half the uniform buffers' worth; a shard holds
            // 1 / RQ_ARENA_SHARDS of it
            uint64_t want = std::max<uint64_t>(idx->arena_hint.load(), (uint64_t)nq * qp.cap / 2);
            unsigned long long total_slots = 0;
            uint32_t arena_rsub = 0;
            bool arena_retried = false;
            ws.arena_failed = true;  // (until the stage has its arena: an allocation failure or a give-up below returns from inside the loop)
#ifdef RQ_DEV_ABLATIONS
            if (g_seg_opt.load() == 3) {  // developer build only (make dev): the arena cannot be had -- through a REAL failing allocation (1 PiB), sticky error and all
                DevBuf<SurvRec> never;
                RQC(never.alloc(1ull << 46));
                return fail(RQ_ERR_OOM, "survivor arena: injected failure (developer hook survivor_segments = 3)");
            }
#endif
            for (int attempt = 0;; ++attempt) {
                want = std::min<uint64_t>(want, 0xFFFF0000ull);
                RQC(ws.arena_recs.ensure(want));
                RQC(ws.arena_runs.ensure(want));
                RQC(ws.arena_cur.ensure(RQ_ARENA_SHARDS + 4));
                RQC(ws.arena_fail.ensure(RQ_ARENA_SHARDS));
                HIPC(hipMemsetAsync(ws.arena_cur.p, 0, (RQ_ARENA_SHARDS + 4) * 8, st));
                HIPC(hipMemsetAsync(ws.arena_fail.p, 0xFF, RQ_ARENA_SHARDS * 4, st));
                ScanExtra hx{};
                RQC(ws.arena_places.ensure(want));
                hx.arena_places = ws.arena_places.p, hx.reserved = nullptr;
                hx.arena_recs = ws.arena_recs.p, hx.arena_runs = reinterpret_cast<uint4 *>(ws.arena_runs.p), hx.arena_cur = ws.arena_cur.p;
                hx.arena_fail = ws.arena_fail.p;
                {  // seven eighths of the arena in shards, the rest as the common area (what a full shard turns away: few, heavy blocks)
                    const uint64_t have = std::min<uint64_t>(ws.arena_recs.count, ws.arena_runs.count);
                    hx.arena_sub = hx.arena_rsub = (uint32_t)(have * 7 / 8 / RQ_ARENA_SHARDS);
                    hx.arena_common = (uint32_t)std::min<uint64_t>(have - (uint64_t)hx.arena_sub * RQ_ARENA_SHARDS, 0xFFFFFF00ull);
                }
                arena_rsub = hx.arena_rsub;
                RQC(ws.scan_extra.ensure(1));
                HIPC(hipMemcpyAsync(ws.scan_extra.p, &hx, sizeof hx, hipMemcpyHostToDevice, st));
                a.x = ws.scan_extra.p;
                a.dense_dir = 0u;
                pf.begin(use_mfma ? PF_SCAN_MATRIX : PF_SCAN);
                if (use_mfma) launch_scan_mfma(sp, a, W, st, additive);  // (the kernel must match the record format stage_fill_kernel wrote)
                else launch_scan(sp, a, W, st);
                pf.end();
                pf.begin(PF_GROUP);
                // sizes from the exact counts, one round trip for the shard-overflow flag and the sum of the segments
                seg_exact_kernel<<<ceil_div(nq, 256), 256, 0, st>>>(ws.surv_cnt.p, nq, 64u, ws.q_cap.p);
                seg_scan_kernel<<<1, 1024, 0, st>>>(ws.q_cap.p, nq, ws.q_base.p, ws.arena_cur.p + RQ_ARENA_SHARDS + 1);
                unsigned long long tail[2] = {0, 0};
                HIPC(hipMemcpyAsync(tail, ws.arena_cur.p + RQ_ARENA_SHARDS, 16, hipMemcpyDeviceToHost, st));
                HIPC(hipStreamSynchronize(st));
                total_slots = tail[1];
                if (!(uint32_t)tail[0]) break;
                pf.end();
                if (attempt >= 6 || want >= 0xFFFF0000ull) return fail(RQ_ERR_OOM, "survivor arena kept overflowing");
                // A shard AND the common area ran full: the stage again (its per-query counters start from zero again) with a
                // larger arena: the exact counts are known now.  A grid of at least 2048 blocks spreads over all the shards: twice
                // the arena, at least the survivors + a quarter.  A SMALL grid uses only a few of the 2048 shards, so doubling
                // alone could stay short for ever (found by the fuzz driver: 700 queries whose every candidate survives, on a
                // 130-block grid): there the common area (an eighth of the arena) is made to hold ALL of the stage's survivors,
                // which takes whatever the shards turn away
                const uint64_t nblocks = a.use_table ? a.ngroups : (uint64_t)a.ngroups * a.tiles_per_group;
                HIPC(hipMemsetAsync(ws.surv_cnt.p, 0, (size_t)nq * sizeof(unsigned long long), st));
                want = std::max<uint64_t>(want * 2, 1u << 20);
                if (nblocks < RQ_ARENA_SHARDS) want = std::max<uint64_t>(want, 8 * total_slots + (1u << 16)), arena_retried = true;
                else want = std::max<uint64_t>(want, total_slots + total_slots / 4);
            }
            ws.arena_failed = false;
            {  // remember what this stage needed
                uint64_t cur = idx->arena_hint.load();
                const uint64_t learnt = std::max<uint64_t>(total_slots + total_slots * 3 / 5, arena_retried ? std::min<uint64_t>(want, 0xFFFF0000ull) : 0ull);
                while (cur < learnt && !const_cast<rq_index *>(idx)->arena_hint.compare_exchange_weak(cur, learnt)) {}
            }
            if (total_slots > ws.surv.count || total_slots > ws.runs.count || total_slots > ws.runs_tmp.count) {
                const uint64_t grow = total_slots + total_slots / 8;
                RQC(ws.surv.ensure(grow));
                RQC(ws.runs.ensure(grow));
                RQC(ws.runs_tmp.ensure(grow));
            }
            sp.surv = ws.surv.p, sp.runs = ws.runs.p;
            arena_scatter_kernel<<<dim3(RQ_ARENA_SHARDS + RQ_ARENA_COMMON_BLOCKS, 2), 256, 0, st>>>(ws.arena_recs.p, reinterpret_cast<const uint4 *>(ws.arena_runs.p), ws.arena_cur.p,
                                                                              ws.arena_fail.p, arena_rsub, ws.q_base.p, ws.arena_places.p, ws.surv.p, ws.runs_tmp.p);
            runs_in_tmp = true;  // the ordering pass below writes the directory
            pf.end();
            seg = QSeg{ws.q_base.p, ws.q_cap.p, qp.cap};
            ws.pend_seg_slots = std::max<uint64_t>(ws.pend_seg_slots, total_slots);
        }
        if (dense_cells) {
            pf.begin(PF_SORT);
            clear_dir_kernel<<<ceil_div((uint64_t)nq * dense_cells, 256), 256, 0, st>>>(ws.runs.p, nq, seg, dense_cells);
            pf.end();
        }
        if (!arena_stage) {
            pf.begin(use_mfma ? PF_SCAN_MATRIX : PF_SCAN);
            if (use_mfma) launch_scan_mfma(sp, a, W, st, additive);
            else launch_scan(sp, a, W, st);
            pf.end();
        }
        if (use_mfma) ws.pend_matrix_stages++;
        if (prof_acc && additive) prof_acc->matrix_additive_launches++;
        if (prof_acc) prof_acc->scan_launches++;
        if (prof_acc && use_mfma) {
            prof_acc->matrix_launches++;
            ws.pend_matrix_ranges.push_back({sg.s_lo, sg.s_hi});
        }
        // small batch: one fused launch per stage (launch-bound regime) -- unless the survivor buffers are large (queries
        // re-run after an overflow: tens of thousands of survivors each): one block per query would rerank and order
        // those alone, the large-batch kernels spread them over the chip
        if (!rq_large_batch(nq) && qp.cap <= 4 * RQ_DEFAULT_CAP) {
            pf.begin(PF_RERANK);
            const uint32_t fin_threads = nq <= 16 ? 1024u : 256u;  // a handful of queries: more lanes on each one's rerank
            // survivor buffers beyond the default mean this index / these queries leave long run directories (overflow
            // re-runs, loose thresholds): those are ordered by the slot-bucketed kernel first; the fused kernel then sorts
            // only what fits its LDS
            const uint32_t presorted = qp.cap > RQ_DEFAULT_CAP && nprobe <= 1024 ? 1u : 0u;
            if (presorted) {
                sort_runs_kernel<<<nq, 64, 0, st>>>(ws.runs.p, ws.surv_cnt.p, seg, ws.big_list.p, ws.big_list.p + nq, RQ_SORT_LDS_RECS, nullptr);
                sort_runs_mid_kernel<<<std::min(nq, 256u), 256, RQ_SORT_MID_LDS_WORDS * 8, st>>>(ws.runs.p, ws.use_runs_tmp ? ws.runs_tmp.p : nullptr, ws.surv_cnt.p,
                                                                                              seg, ws.big_list.p, ws.big_list.p + nq, nprobe, 0u,
                                                                                              RQ_SORT_MID_LDS_WORDS);
            }
            if (sb_fused_finish) {  // small-batch path, heap ranker: the stage's finish also writes the results and the totals
                // a handful of queries: their final-stage survivors (~1000 rows each) are gathered by the whole chip -- one block
                // per query would pull them through a single CU's memory pipeline (~30 GB/s)
                uint32_t flags = presorted;
                if (nq <= 32) {
                    accurate_kernel<<<dim3(std::max(1u, std::min(16u, 256u / nq)), nq), 256, (size_t)dim * sizeof(float), st>>>(
                        ws.surv.p, ws.surv_cnt.p, seg, idx->view(), qpad, dim, nullptr, probe_cluster, nprobe);
                    flags |= 2u;
                }
                const uint32_t presorted = flags;
                if (topk < 64)
                    sb_finish_kernel<true><<<nq, fin_threads, (size_t)dim * sizeof(float) + (2 * RQ_SBF_RUNS + RQ_SBF_RECS) * 16, st>>>(
                        ws.surv.p, ws.runs.p, ws.surv_cnt.p, seg, idx->view(), qpad, dim, topk, rs, probe_cluster, nprobe, presorted,
                        idx->map_ids.p, d_out_dist, d_out_id, d_out_n, ws.rough_cnt.p, ws.totals.p);
                else
                    sb_finish_kernel<false><<<nq, fin_threads, (size_t)dim * sizeof(float) + (2 * RQ_SBF_RUNS + RQ_SBF_RECS) * 16, st>>>(
                        ws.surv.p, ws.runs.p, ws.surv_cnt.p, seg, idx->view(), qpad, dim, topk, rs, probe_cluster, nprobe, presorted,
                        idx->map_ids.p, d_out_dist, d_out_id, d_out_n, ws.rough_cnt.p, ws.totals.p);
                sb_results_done = true;
            } else if (qp.heuristic)
                stage_finish_kernel<true><<<nq, fin_threads, (size_t)dim * sizeof(float), st>>>(ws.surv.p, ws.runs.p, ws.surv_cnt.p, seg, idx->view(),
                                                              qpad, dim, topk, rs, probe_cluster, nprobe, presorted);
            else
                stage_finish_kernel<false><<<nq, fin_threads, (size_t)dim * sizeof(float), st>>>(ws.surv.p, ws.runs.p, ws.surv_cnt.p, seg, idx->view(),
                                                               qpad, dim, topk, rs, probe_cluster, nprobe, presorted);
            pf.end();
        } else {  // large batch: full-chip rerank, then run-directory sort, then one replay wave per query
            pf.begin(PF_RERANK);
            const uint32_t gx = std::max(1u, std::min(16u, 4096u / std::max(nq, 1u)));
            // past the first stage the thresholds are finite: survivors go through the fp16 shadow rows first
            if (idx->base_q8.p && (stage_no > 0 || qp.thr_init) && !(g_scan_dbg & 512))
                accurate_filtered8_kernel<<<dim3(gx, nq), 256, (size_t)dim * sizeof(float) + (nprobe <= RQ_ACC8_LDS_PROBES ? (size_t)nprobe * 16 : 0), st>>>(
                    ws.surv.p, ws.surv_cnt.p, seg, idx->base.p, idx->base_q8.p, idx->list_q8.p, qpad, dim, rerank_order, ws.thr.p,
                    probe_cluster, nprobe, ws.nshadow.p, k);
            else if (idx->base_h.p && (stage_no > 0 || qp.thr_init) && !(g_scan_dbg & 512))
                accurate_filtered_kernel<<<dim3(gx, nq), 256, (size_t)dim * sizeof(float), st>>>(
                    ws.surv.p, ws.surv_cnt.p, seg, idx->base.p, idx->base_h.p, qpad, dim, rerank_order, ws.thr.p,
                    ws.nshadow.p);
            else if (idx->split_rows && (stage_no > 0 || qp.thr_init) && !(g_scan_dbg & 512))  // tiered: the rows' own first plane is the pre-filter
                accurate_split_kernel<<<dim3(gx, nq), 256, (size_t)dim * sizeof(float) + (nprobe <= RQ_ACC8_LDS_PROBES ? (size_t)nprobe * 16 : 0), st>>>(
                    ws.surv.p, ws.surv_cnt.p, seg, idx->view(), qpad, dim, rerank_order, ws.thr.p, probe_cluster, nprobe, ws.nshadow.p);
            else
                accurate_kernel<<<dim3(gx, nq), 256, (size_t)dim * sizeof(float), st>>>(ws.surv.p, ws.surv_cnt.p, seg, idx->view(), qpad, dim,
                                                                                        rerank_order, probe_cluster, nprobe);
            pf.end();
            if (!dense_cells) {
                pf.begin(PF_SORT);
                sort_runs_kernel<<<nq, 64, 0, st>>>(ws.runs.p, ws.surv_cnt.p, seg, ws.big_list.p, ws.big_list.p + nq, 512u,
                                                    runs_in_tmp ? ws.runs_tmp.p : nullptr);
                // queries with long run directories (loose thresholds, very unequal lists): cell-bitmap ordering, persistent blocks walking the list
                sort_runs_mid_kernel<<<mid_blocks, 256, RQ_SORT_MID_LDS_WORDS * 8, st>>>(ws.runs.p, (ws.use_runs_tmp || runs_in_tmp) ? ws.runs_tmp.p : nullptr,
                                                                                      ws.surv_cnt.p, seg, ws.big_list.p, ws.big_list.p + nq, nprobe,
                                                                                      runs_in_tmp ? 1u : 0u, RQ_SORT_MID_LDS_WORDS);
                pf.end();
            }
            pf.begin(PF_REPLAY);
            if (qp.heuristic)
                replay_kernel<true><<<nq, 64, 16, st>>>(ws.surv.p, ws.runs.p, ws.surv_cnt.p, seg, topk, rs, dense_cells);
            else if (topk < 64)  // the heap in registers, one element per lane (a push before a pop holds topk + 1 elements)
                replay_kernel<false, true><<<nq, 64, 16, st>>>(ws.surv.p, ws.runs.p, ws.surv_cnt.p, seg, topk, rs, dense_cells);
            else
                replay_kernel<false><<<nq, 64, (size_t)topk * 8, st>>>(ws.surv.p, ws.runs.p, ws.surv_cnt.p, seg, topk, rs, dense_cells);
            pf.end();
        }
    }

    // 6. results
    pf.begin(PF_REPLAY);
    if (sb_results_done) {
        // written by sb_query_kernel / sb_finish_kernel together with the totals
    } else {
    if (qp.heuristic) {
        sort_survivors_kernel<<<nq, 256, 0, st>>>(ws.arr.p, ws.arr_len.p, qp.hcap);
        finalize_heuristic_kernel<<<ceil_div((uint64_t)nq * topk, 256), 256, 0, st>>>(rs, nq, topk, d_row_map, idx->map_ids.p,
                                                                                       d_out_dist, d_out_id, d_out_n);
    } else {
        finalize_heap_kernel<<<ceil_div((uint64_t)nq * topk, 256), 256, 0, st>>>(rs, nq, topk, d_row_map, idx->map_ids.p, d_out_dist,
                                                                                  d_out_id, d_out_n);
    }
    metrics_sum_kernel<<<std::min(256u, ceil_div(nq, 256)), 256, 0, st>>>(
        ws.rough_cnt.p, ws.precise.p, ws.need.p, qp.heuristic ? ws.arr_len.p : nullptr, ws.nsurv.p, ws.nshadow.p, nq, ws.ovf.p, qp.hcap,
        ws.totals.p);
    }
    pf.end();
    if (pf.on) (void)hipEventRecord(pf.spans[total_span].b, st);
    HIPC(hipMemcpyAsync(ws.h_totals, ws.totals.p, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    ws.h_totals[7] = 0, ws.h_totals[8] = 0, ws.h_totals[9] = 0, ws.h_totals[10] = 0;
    if (ws.pend_prefiltered) HIPC(hipMemcpyAsync(ws.h_totals + 10, ws.totals.p + 12, 8, hipMemcpyDeviceToHost, st));
    if (ws.pend_matrix_stages) {  // sub-tile steps of the matrix-core stages and how many of them took the exact path
        stat_fold_kernel<<<1, 64, 0, st>>>(ws.stat.p, ws.stat.p + 200);
        HIPC(hipMemcpyAsync(ws.h_totals + 8, ws.stat.p + 200, 16, hipMemcpyDeviceToHost, st));
    }
    if (rq_large_batch(nq))  // (the long-directory hint only sizes launches of large batches: a small batch saves the copy's round trip)
        HIPC(hipMemcpyAsync(ws.h_totals + 7, ws.big_list.p + nq + 2, 4, hipMemcpyDeviceToHost, st));
    ws.pend_total_span = total_span;
    ws.pend_nq = nq;
    ws.pend_cap = qp.cap;
    if (defer) return RQ_OK;  // the caller finishes the pass later (rq_query_batch_device_end)
    return finish_pass(idx, ws, res, prof_acc);
}

static Workspace *ws_acquire(rq_index *idx) {
    std::lock_guard<std::mutex> g(idx->ws_mu);
    for (auto &w : idx->ws_pool)
        if (!w->busy) {
            w->busy = true;
            return w.get();
        }
    idx->ws_pool.emplace_back(new Workspace());
    idx->ws_pool.back()->busy = true;
    return idx->ws_pool.back().get();
}
static void ws_release(rq_index *idx, Workspace *w) {
    std::lock_guard<std::mutex> g(idx->ws_mu);
    w->busy = false;
}

static rq_status validate_query(const rq_index *idx, const float *d_q, uint32_t len, uint32_t probe, uint32_t topk,
                                const float *d_out_dist, const uint32_t *d_out_id, const uint32_t *d_out_n) {
    RQC(ensure_device());
    RQC(ensure_kernel_attributes());
    if (!idx || !d_q || !d_out_dist || !d_out_id || !d_out_n) return fail(RQ_ERR_INVALID, "null argument");
    if (idx->dim != (len + 63) / 64 * 64)  // rabitq.rs:275
        return fail(RQ_ERR_DIM_MISMATCH, "query length " + std::to_string(len) + " does not pad to index dim " +
                                             std::to_string(idx->dim));
    if (probe == 0 || idx->k == 0) return fail(RQ_ERR_INVALID, "probe == 0 (the reference panics at rabitq.rs:295)");
    if (topk == 0 || topk > RQ_MAX_TOPK) return fail(RQ_ERR_UNSUPPORTED, "topk must be in [1, 2048]");
    if (std::min(probe, idx->k) > RQ_MAX_PROBE) return fail(RQ_ERR_UNSUPPORTED, "probe > 16384 not supported");
    if (idx->dim > 4096) return fail(RQ_ERR_UNSUPPORTED, "dim > 4096 not supported");
    return RQ_OK;
}

// Uniform survivor capacity of a pass over `remaining` queries, and whether its final stage is segmented.  An index whose
// batches overflowed the default capacity (cap_hint) used to size EVERY query of a pass for the worst one (learnt capacity
// 32 768: 100 GB for a 65 536-query pass of the hard benchmark distribution); large batches now keep the default
// capacity for the stages that cannot exceed it and give every other stage per-query segments.
static uint32_t pass_capacity(const rq_index *idx, uint32_t remaining, bool seeded, bool *seg) {
    const uint32_t hint = idx->cap_hint.load();
    const int opt = g_seg_opt.load();
    *seg = !seeded && rq_large_batch(remaining) && scan_is_fused(idx->W) && (opt >= 2 || (opt == 1 && hint > RQ_DEFAULT_CAP));
    if (*seg) return RQ_DEFAULT_CAP;  // stages that cannot exceed it stay uniform, the others are segmented
    return std::max(RQ_DEFAULT_CAP, hint);
}

// queries per pass: survivor / run buffers are 32 B per slot per query (keep one pass under ~24 GiB) and
// (query, list) pairs per pass <= 2^22 (bounds the per-pair buffers and every launch size)
static uint32_t pass_queries(const rq_index *idx, uint32_t remaining, uint32_t probe, uint32_t cap0, bool seg, bool ext_lists = false) {
    // survivor records + run directory: 32 B per slot per query, 48 B when the pass also keeps the second directory buffer
    // (ws_prepare: capacities beyond the default, segmented passes, long directories); the budget is a third of the HBM that
    // was free once the index was resident (at least 4 GiB: an index that fills the HBM -- 100M x 768 -- still answers a
    // 32 768-query batch in ONE pass; split in two, every block of the matrix-core scan paid its start-up twice: a third
    // of that launch at dim 768)
    const uint64_t slot_bytes = cap0 > RQ_DEFAULT_CAP || seg || idx->big_dirs_hint.load() > 0 ? 48 : 32;
    // Probe lists supplied by the caller = a shard of a multi-GPU deployment: most of a query's probed lists live on other ranks
    // (empty here: skipped before any per-pair work), and the step's batch grows with the number of ranks so that a list still
    // meets as many queries as on one GPU -- cut into passes of 65 536 queries, each pass of an 8-GPU step would bring a list
    // 128 queries instead of 1024 and the matrix-core scan would run at half its rate (one-rank-of-eight rehearsal: 0.19 of
    // peak).  Such passes may hold 16 x the queries / pairs (per-pair buffers: ~200 B per pair, 6.7 GB at 2^25 pairs).
    const uint64_t max_nq = ext_lists ? 16ull * RQ_MAX_NQ_PER_PASS : RQ_MAX_NQ_PER_PASS, max_pairs = ext_lists ? (1ull << 26) : (1ull << 22);
    uint32_t step_nq = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(remaining, max_nq),
                                                    std::max<uint64_t>(1, idx->pass_budget / ((uint64_t)cap0 * slot_bytes)));
    return std::min<uint32_t>(step_nq, (uint32_t)std::max<uint64_t>(1, max_pairs / std::min(probe, idx->k)));
}

// After a finished pass: remember the capacity it needed and re-run exactly the queries whose survivor
// buffers overflowed, with the capacity they asked for.  All pointers are those of the pass (already offset).
static rq_status after_pass(rq_index *idx, Workspace *ws, const QueryParams &qp, const float *d_q, float *d_out_dist,
                            uint32_t *d_out_id, uint32_t *d_out_n, const uint32_t *ext_cluster, const float *ext_dist,
                            const PassResult &pr, rq_profile_t &prof, uint64_t &tot_precise) {
    const uint32_t len = qp.len, probe = qp.probe, topk = qp.topk;
    const bool heuristic = qp.heuristic;
    const uint32_t npb = std::min(probe, idx->k);
    if (pr.max_need > qp.cap) {  // remember (with headroom) so that later batches do not overflow
        // ... but only up to RQ_MAX_CAP_HINT: survivor buffers are cap x 32 B for EVERY query of the pass, so one outlier
        // query (a loose threshold after an unlucky nearest list) must not shrink the passes of all later batches; beyond
        // the bound the outliers are simply re-run below with the capacity they asked for
        uint32_t want = pow2_ceil((uint32_t)std::min<uint64_t>(pr.max_need + pr.max_need / 4, RQ_MAX_CAP_HINT));
        uint32_t cur = idx->cap_hint.load();
        while (cur < want && !idx->cap_hint.compare_exchange_weak(cur, want)) {}
    }
    if (!pr.overflowed) return RQ_OK;
    uint32_t cap = qp.cap, hcap = qp.hcap;
    std::vector<uint32_t> h_need(qp.nq), h_alen(qp.nq), h_ovf(qp.nq), over_rows;
    HIPC(hipMemcpy(h_need.data(), ws->need.p, qp.nq * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(h_alen.data(), ws->arr_len.p, qp.nq * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(h_ovf.data(), ws->ovf.p, qp.nq * 4, hipMemcpyDeviceToHost));
    uint32_t max_need = 0, max_alen = 0;
    for (uint32_t b = 0; b < qp.nq; ++b)
        if (h_ovf[b] || (heuristic && h_alen[b] > hcap)) {
            over_rows.push_back(b);
            max_need = std::max(max_need, h_need[b]);
            max_alen = std::max(max_alen, h_alen[b]);
        }
    int guard = 0;
    while (!over_rows.empty() && guard++ < 8) {
        prof.retries += (uint32_t)over_rows.size();
        uint32_t ncap = std::max(cap * 2, pow2_ceil(max_need));
        uint32_t nhcap = heuristic ? std::max(hcap * 2, pow2_ceil(std::max(max_alen, max_need))) : hcap;
        // bound the retry workspace to ~4 GiB of survivor records
        uint32_t chunk = (uint32_t)std::max<uint64_t>(1, (4ull << 30) / ((uint64_t)(ncap + nhcap) * sizeof(SurvRec)));
        std::vector<uint32_t> still;
        // a pooled workspace (its buffers persist: a workload whose outliers overflow every batch must not pay
        // hipMalloc / hipFree of gigabytes per batch)
        Workspace *rwsp = ws_acquire(idx);
        struct RelR {
            rq_index *i;
            Workspace *w;
            ~RelR() { ws_release(i, w); }
        } relr{idx, rwsp};
        Workspace &rws = *rwsp;
        DevBuf<float> &sub_q = rws.retry_q;
        DevBuf<uint32_t> &sub_rows = rws.retry_rows;
        for (size_t o = 0; o < over_rows.size(); o += chunk) {
            uint32_t m = (uint32_t)std::min<size_t>(chunk, over_rows.size() - o);
            QueryParams rq{m, len, probe, topk, heuristic, ncap, nhcap};
            rq.thr_init = qp.thr_init;  // indexed through the row map
            RQC(ws_prepare(idx, rws, rq));
            RQC(sub_q.ensure((uint64_t)m * len));
            RQC(sub_rows.ensure(m));
            HIPC(hipMemcpy(sub_rows.p, over_rows.data() + o, m * 4, hipMemcpyHostToDevice));
            gather_rows_kernel<<<ceil_div((uint64_t)m * len, 256), 256, 0, rws.stream>>>(d_q, sub_rows.p, m, len, sub_q.p);
            PassResult rr;
            const uint32_t *sub_pc = nullptr;
            const float *sub_pd = nullptr;
            DevBuf<float> &sub_probe_d = rws.retry_pd, &sub_probe_c = rws.retry_pc;
            if (ext_cluster) {  // the caller's probe lists, restricted to the re-run queries
                RQC(sub_probe_c.ensure((uint64_t)m * npb));
                RQC(sub_probe_d.ensure((uint64_t)m * npb));
                gather_rows_kernel<<<ceil_div((uint64_t)m * npb, 256), 256, 0, rws.stream>>>(
                    reinterpret_cast<const float *>(ext_cluster), sub_rows.p, m, npb, sub_probe_c.p);
                gather_rows_kernel<<<ceil_div((uint64_t)m * npb, 256), 256, 0, rws.stream>>>(ext_dist, sub_rows.p, m, npb,
                                                                                              sub_probe_d.p);
                sub_pc = reinterpret_cast<const uint32_t *>(sub_probe_c.p);
                sub_pd = sub_probe_d.p;
            }
            RQC(run_pass(idx, rws, sub_q.p, rq, sub_rows.p, d_out_dist, d_out_id, d_out_n, &rr, nullptr, sub_pc, sub_pd));
            tot_precise += rr.precise;
            if (rr.overflowed) {
                std::vector<uint32_t> n2(m), a2(m), o2(m);
                HIPC(hipMemcpy(n2.data(), rws.need.p, m * 4, hipMemcpyDeviceToHost));
                HIPC(hipMemcpy(a2.data(), rws.arr_len.p, m * 4, hipMemcpyDeviceToHost));
                HIPC(hipMemcpy(o2.data(), rws.ovf.p, m * 4, hipMemcpyDeviceToHost));
                for (uint32_t b = 0; b < m; ++b)
                    if (o2[b] || (heuristic && a2[b] > nhcap)) {
                        still.push_back(over_rows[o + b]);
                        max_need = std::max(max_need, n2[b]);
                        max_alen = std::max(max_alen, a2[b]);
                    }
            }
        }
        cap = ncap, hcap = nhcap;
        over_rows.swap(still);
    }
    if (!over_rows.empty()) return fail(RQ_ERR_OOM, "survivor buffers kept overflowing");
    return RQ_OK;
}

// tail of every query call: the reference's panics and counters
static rq_status conclude_query(uint32_t nq, bool heuristic, const uint32_t *d_out_n, uint64_t tot_rough,
                                uint64_t tot_precise, const rq_profile_t &prof) {
    bool any_empty = false;
    if (heuristic) {  // rerank.rs:171-173: an empty array panics in the reference
        std::vector<uint32_t> h_n(nq);
        HIPC(hipMemcpy(h_n.data(), d_out_n, nq * 4, hipMemcpyDefault));  // d_out_n may be device or mapped host memory
        for (uint32_t v : h_n) any_empty |= (v == 0);
    }
    g_rough.fetch_add(tot_rough, std::memory_order_relaxed);      // rerank.rs:105
    g_precise.fetch_add(tot_precise, std::memory_order_relaxed);  // rerank.rs:104
    g_query.fetch_add(nq, std::memory_order_relaxed);             // rabitq.rs:331
    g_profile = prof;
    if (any_empty) return fail(RQ_ERR_EMPTY, "heuristic ranker accepted no candidate for at least one query");
    return RQ_OK;
}

// queries/outputs in device memory
static rq_status query_device(rq_index *idx, const float *d_q, uint32_t nq, uint32_t len, uint32_t probe,
                              uint32_t topk, bool heuristic, float *d_out_dist, uint32_t *d_out_id,
                              uint32_t *d_out_n, const uint32_t *ext_cluster = nullptr,
                              const float *ext_dist = nullptr, Workspace *use_ws = nullptr, const float *ext_thr = nullptr) {
    RQC(validate_query(idx, d_q, len, probe, topk, d_out_dist, d_out_id, d_out_n));
    if (nq == 0) return RQ_OK;
    rq_profile_t prof;
    memset(&prof, 0, sizeof prof);
    Workspace *ws = use_ws ? use_ws : ws_acquire(idx);  // use_ws: the caller holds (and releases) the workspace
    struct Rel {
        rq_index *i;
        Workspace *w;
        ~Rel() {
            if (w) ws_release(i, w);
        }
    } rel{idx, use_ws ? nullptr : ws};
    uint64_t tot_rough = 0, tot_precise = 0;
    const uint32_t npb = std::min(probe, idx->k);
    // A call of several passes (more than 65 536 queries) keeps TWO passes in flight, each on a workspace and stream of its own -- what
    // rq_query_batch_device_begin / _end let a caller do with two batches, done here for one large batch: pass i + 1 is enqueued
    // before pass i is waited for, so the thin launches at either end of a pass overlap the other pass's wide ones (+4 %: 131 072
    // queries 33.9 -> 32.6 ms per call).  Results are those of the passes run one after the other.
    if (!use_ws && !ext_cluster && !ext_thr && g_pass_overlap.load()) {
        bool seg0 = false;
        const uint32_t cap_first = pass_capacity(idx, nq, false, &seg0);
        const uint32_t first_nq = pass_queries(idx, nq, probe, cap_first, seg0);
        // (only with room for a second workspace: an index that fills the HBM -- configs[3] -- runs its passes one after the other;
        // rough size of a pass's buffers: survivor records + directories, per-pair records and operands, distances, ranker state)
        bool room = false;
        if (first_nq < nq) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const uint64_t need = (uint64_t)first_nq * ((uint64_t)cap_first * 48 + (uint64_t)npb * 320 + (uint64_t)idx->dim * 8 + (uint64_t)idx->k * 4 + 4096);
                room = free_b > need + (6ull << 30);
            }
        }
        if (first_nq < nq && room) {
            ws_release(idx, ws);  // (the passes take workspaces of their own from the pool, this one among them)
            rel.w = nullptr;
            struct Flight {
                Workspace *ws = nullptr;
                QueryParams qp{};
                uint32_t q0 = 0;
            } fl[2];
            auto finish = [&](Flight &f) -> rq_status {
                if (!f.ws) return RQ_OK;
                Workspace *w = f.ws;
                f.ws = nullptr;
                struct R {
                    rq_index *i;
                    Workspace *w;
                    ~R() { ws_release(i, w); }
                } r{idx, w};
                PassResult pr;
                RQC(finish_pass(idx, *w, &pr, &prof));
                tot_rough += pr.rough;
                tot_precise += pr.precise;
                return after_pass(idx, w, f.qp, d_q + (uint64_t)f.q0 * len, d_out_dist + (uint64_t)f.q0 * topk, d_out_id + (uint64_t)f.q0 * topk,
                                  d_out_n + f.q0, nullptr, nullptr, pr, prof, tot_precise);
            };
            auto drain = [&]() {  // (an error path: nothing may stay in flight, every workspace goes back)
                for (Flight &f : fl)
                    if (f.ws) {
                        (void)hipStreamSynchronize(f.ws->stream);
                        ws_release(idx, f.ws);
                        f.ws = nullptr;
                    }
            };
            uint32_t slot = 0;
            for (uint32_t q0 = 0, step_nq = 0; q0 < nq; q0 += step_nq, slot ^= 1u) {
                rq_status st = finish(fl[slot]);  // the pass before the previous one
                bool seg = false;
                const uint32_t cap0 = pass_capacity(idx, nq - q0, false, &seg);
                step_nq = pass_queries(idx, nq - q0, probe, cap0, seg);
                Flight &f = fl[slot];
                if (st == RQ_OK) {
                    f.qp = QueryParams{step_nq, len, probe, topk, heuristic, cap0, std::max(cap0, std::max(RQ_DEFAULT_CAP, idx->cap_hint.load()))};
                    f.qp.seg_final = seg && rq_large_batch(step_nq);
                    f.q0 = q0;
                    f.ws = ws_acquire(idx);
                    st = ws_prepare(idx, *f.ws, f.qp);
                }
                if (st == RQ_OK) {
                    f.ws->arena_failed = false;
                    PassResult pr;
                    st = run_pass(idx, *f.ws, d_q + (uint64_t)q0 * len, f.qp, nullptr, d_out_dist + (uint64_t)q0 * topk, d_out_id + (uint64_t)q0 * topk,
                                  d_out_n + q0, &pr, &prof, nullptr, nullptr, true);
                    if (st != RQ_OK && f.ws->arena_failed && f.qp.seg_final) {  // as below: the pass again on the uniform buffers
                        (void)hipStreamSynchronize(f.ws->stream);
                        (void)hipGetLastError();
                        f.ws->arena_recs.release(), f.ws->arena_runs.release(), f.ws->scan_extra.release(), f.ws->arena_places.release();
                        f.ws->arena_failed = false;
                        f.qp.seg_final = false;
                        st = ws_prepare(idx, *f.ws, f.qp);
                        if (st == RQ_OK)
                            st = run_pass(idx, *f.ws, d_q + (uint64_t)q0 * len, f.qp, nullptr, d_out_dist + (uint64_t)q0 * topk,
                                          d_out_id + (uint64_t)q0 * topk, d_out_n + q0, &pr, &prof, nullptr, nullptr, true);
                    }
                }
                if (st != RQ_OK) {
                    drain();
                    return st;
                }
            }
            for (uint32_t i = 0; i < 2; ++i, slot ^= 1u) {  // oldest first
                const rq_status st = finish(fl[slot]);
                if (st != RQ_OK) {
                    drain();
                    return st;
                }
            }
            return conclude_query(nq, heuristic, d_out_n, tot_rough, tot_precise, prof);
        }
    }
    for (uint32_t q0 = 0, step_nq = 0; q0 < nq; q0 += step_nq) {
        bool seg = false;
        const uint32_t cap0 = pass_capacity(idx, nq - q0, ext_thr != nullptr, &seg);
        step_nq = pass_queries(idx, nq - q0, probe, cap0, seg, ext_cluster != nullptr);
        QueryParams qp{step_nq, len, probe, topk, heuristic, cap0, std::max(cap0, std::max(RQ_DEFAULT_CAP, idx->cap_hint.load()))};
        qp.seg_final = seg && rq_large_batch(step_nq);
        qp.thr_init = ext_thr ? ext_thr + q0 : nullptr;
        qp.ext_lists = ext_cluster != nullptr;
        RQC(ws_prepare(idx, *ws, qp));
        PassResult pr;
        const float *q_at = d_q + (uint64_t)q0 * len;
        float *od = d_out_dist + (uint64_t)q0 * topk;
        uint32_t *oi = d_out_id + (uint64_t)q0 * topk, *on = d_out_n + q0;
        const uint32_t *ec = ext_cluster ? ext_cluster + (uint64_t)q0 * npb : nullptr;
        const float *ed = ext_dist ? ext_dist + (uint64_t)q0 * npb : nullptr;
        ws->arena_failed = false;
        rq_status ps = run_pass(idx, *ws, q_at, qp, nullptr, od, oi, on, &pr, &prof, ec, ed);
        if (ps != RQ_OK && ws->arena_failed && qp.seg_final) {
            // no room for the survivor arena (or it kept overflowing): the pass again on the uniform buffers, where a query that
            // overflows is simply re-run with the capacity it asks for -- slower, never wrong
            (void)hipStreamSynchronize(ws->stream);
            // a failed hipMalloc leaves hipErrorOutOfMemory as the thread's last error (sticky on ROCm 7.2): the repeat's own
            // hipGetLastError() check must not pick it up; the arena of earlier batches goes back to the pool the repeat allocates from
            (void)hipGetLastError();
            ws->arena_recs.release(), ws->arena_runs.release(), ws->scan_extra.release(), ws->arena_places.release();
            ws->arena_failed = false;
            qp.seg_final = false;
            RQC(ws_prepare(idx, *ws, qp));
            pr = PassResult();
            ps = run_pass(idx, *ws, q_at, qp, nullptr, od, oi, on, &pr, &prof, ec, ed);
        }
        RQC(ps);
        tot_rough += pr.rough;
        tot_precise += pr.precise;
        RQC(after_pass(idx, ws, qp, q_at, od, oi, on, ec, ed, pr, prof, tot_precise));
    }
    return conclude_query(nq, heuristic, d_out_n, tot_rough, tot_precise, prof);
}

// The same call split in two, so that a caller can keep several batches in flight (each on its own
// workspace and HIP stream): begin enqueues the whole pass and returns, end waits for it and does the
// (rare) overflow re-runs.  Calls that need more than one pass run synchronously inside begin.
struct rq_ticket {
    rq_index *idx = nullptr;
    Workspace *ws = nullptr;
    QueryParams qp{};
    const float *d_q = nullptr;
    float *d_out_dist = nullptr;
    uint32_t *d_out_id = nullptr, *d_out_n = nullptr;
    const uint32_t *ext_cluster = nullptr;
    const float *ext_dist = nullptr;
    rq_profile_t prof;
    bool done = false;
    rq_status status = RQ_OK;
};

static rq_status query_device_begin(rq_index *idx, const float *d_q, uint32_t nq, uint32_t len, uint32_t probe,
                                    uint32_t topk, bool heuristic, float *d_out_dist, uint32_t *d_out_id,
                                    uint32_t *d_out_n, rq_ticket **out) {
    if (!out) return fail(RQ_ERR_INVALID, "null argument");
    *out = nullptr;
    RQC(validate_query(idx, d_q, len, probe, topk, d_out_dist, d_out_id, d_out_n));
    std::unique_ptr<rq_ticket> t(new rq_ticket());
    t->idx = idx;
    memset(&t->prof, 0, sizeof t->prof);
    bool seg = false;
    const uint32_t cap0 = pass_capacity(idx, nq, false, &seg);
    if (nq == 0 || pass_queries(idx, nq, probe, cap0, seg) < nq) {  // nothing to overlap / several passes: synchronous
        t->status = query_device(idx, d_q, nq, len, probe, topk, heuristic, d_out_dist, d_out_id, d_out_n);
        t->done = true;
        *out = t.release();
        return RQ_OK;
    }
    t->qp = QueryParams{nq, len, probe, topk, heuristic, cap0, std::max(cap0, std::max(RQ_DEFAULT_CAP, idx->cap_hint.load()))};
    t->qp.seg_final = seg && rq_large_batch(nq);
    t->d_q = d_q, t->d_out_dist = d_out_dist, t->d_out_id = d_out_id, t->d_out_n = d_out_n;
    t->ws = ws_acquire(idx);
    rq_status st = ws_prepare(idx, *t->ws, t->qp);
    PassResult pr;
    if (st == RQ_OK) {
        t->ws->arena_failed = false;
        st = run_pass(idx, *t->ws, d_q, t->qp, nullptr, d_out_dist, d_out_id, d_out_n, &pr, &t->prof, nullptr, nullptr, true);
        if (st != RQ_OK && t->ws->arena_failed && t->qp.seg_final) {  // as in query_device: the pass again on the uniform buffers
            (void)hipStreamSynchronize(t->ws->stream);
            (void)hipGetLastError();  // (a failed hipMalloc's sticky error, as in query_device)
            t->ws->arena_recs.release(), t->ws->arena_runs.release(), t->ws->scan_extra.release(), t->ws->arena_places.release();
            t->ws->arena_failed = false;
            t->qp.seg_final = false;
            st = ws_prepare(idx, *t->ws, t->qp);
            if (st == RQ_OK) st = run_pass(idx, *t->ws, d_q, t->qp, nullptr, d_out_dist, d_out_id, d_out_n, &pr, &t->prof, nullptr, nullptr, true);
        }
    }
    if (st != RQ_OK) {
        (void)hipStreamSynchronize(t->ws->stream);
        ws_release(idx, t->ws);
        return st;
    }
    *out = t.release();
    return RQ_OK;
}

static rq_status query_device_end(rq_ticket *tk) {
    if (!tk) return fail(RQ_ERR_INVALID, "null ticket");
    std::unique_ptr<rq_ticket> t(tk);
    if (t->done) return t->status;
    struct Rel {
        rq_index *i;
        Workspace *w;
        ~Rel() { ws_release(i, w); }
    } rel{t->idx, t->ws};
    PassResult pr;
    RQC(finish_pass(t->idx, *t->ws, &pr, &t->prof));
    uint64_t tot_precise = pr.precise;
    RQC(after_pass(t->idx, t->ws, t->qp, t->d_q, t->d_out_dist, t->d_out_id, t->d_out_n, nullptr, nullptr, pr, t->prof,
                   tot_precise));
    return conclude_query(t->qp.nq, t->qp.heuristic, t->d_out_n, pr.rough, tot_precise, t->prof);
}

