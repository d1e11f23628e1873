// host_shard.h -- part of the host side of librabitq_hip.so (one translation unit: rabitq_hip.hip includes the host_*.h files in order;
// they are not stand-alone headers).  Multi-GPU (inside extern "C"): list partitioner, shard carving, the sharded step over an RCCL communicator (or caller-supplied collectives).
#pragma once
// ---- multi-GPU: list partitioner, shard carving, the sharded step over an RCCL communicator ------------------
rq_status rq_partition_lists(const rq_index *idx, uint32_t world, uint32_t *out_owner, uint64_t *out_load) {
    if (!idx || !out_owner || world == 0) return fail(RQ_ERR_INVALID, "bad partition arguments");
    // whole lists to shards, greedy by list length (longest first, each to the least-loaded shard; ties: lower list
    // id first, lower shard first): deterministic, so every rank computes the same assignment on its own
    std::vector<uint32_t> off((size_t)idx->k + 1);
    RQC(rq_get_array(idx, RQ_ARR_OFFSETS, off.data(), off.size() * 4));
    std::vector<uint32_t> order(idx->k);
    for (uint32_t c = 0; c < idx->k; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(),
                     [&](uint32_t a, uint32_t b) { return off[a + 1] - off[a] > off[b + 1] - off[b]; });
    typedef std::pair<uint64_t, uint32_t> LS;  // (load, shard): min-heap
    std::priority_queue<LS, std::vector<LS>, std::greater<LS>> heap;
    for (uint32_t r = 0; r < world; ++r) heap.push({0, r});
    for (uint32_t c : order) {
        LS t = heap.top();
        heap.pop();
        out_owner[c] = t.second;
        t.first += off[c + 1] - off[c];
        heap.push(t);
    }
    if (out_load) {
        for (uint32_t r = 0; r < world; ++r) out_load[r] = 0;
        for (uint32_t c = 0; c < idx->k; ++c) out_load[out_owner[c]] += off[c + 1] - off[c];
    }
    return RQ_OK;
}

rq_status rq_shard_index(const rq_index *idx, const uint32_t *owner, uint32_t rank, rq_index **out) {
    RQC(ensure_device());
    RQC(ensure_kernel_attributes());
    if (!idx || !owner || !out) return fail(RQ_ERR_INVALID, "null argument");
    *out = nullptr;
    const uint32_t k = idx->k, dim = idx->dim;
    std::vector<uint32_t> off((size_t)k + 1), noff((size_t)k + 1);
    RQC(rq_get_array(idx, RQ_ARR_OFFSETS, off.data(), off.size() * 4));
    uint64_t n_local = 0;
    for (uint32_t c = 0; c < k; ++c) {
        noff[c] = (uint32_t)n_local;
        if (owner[c] == rank) n_local += off[c + 1] - off[c];
    }
    noff[k] = (uint32_t)n_local;
    std::unique_ptr<rq_index> sh(new rq_index());
    sh->dim = dim, sh->k = k, sh->n = n_local, sh->W = idx->W;
    RQC(sh->P.alloc((size_t)dim * dim));
    RQC(sh->centroids.alloc((size_t)k * dim));
    RQC(sh->offsets.alloc((size_t)k + 1));
    RQC(alloc_base_tiers(sh.get(), 0, noff.data()));
    RQC(sh->codes.alloc(n_local * sh->W));
    RQC(sh->factors.alloc(n_local));
    RQC(sh->map_ids.alloc(n_local));
    HIPC(hipMemcpy(sh->P.p, idx->P.p, (size_t)dim * dim * 4, hipMemcpyDeviceToDevice));
    HIPC(hipMemcpy(sh->centroids.p, idx->centroids.p, (size_t)k * dim * 4, hipMemcpyDeviceToDevice));  // all centroids replicated
    HIPC(hipMemcpy(sh->offsets.p, noff.data(), ((size_t)k + 1) * 4, hipMemcpyHostToDevice));
    if (n_local)
        shard_gather_kernel<<<(uint32_t)std::min<uint64_t>(ceil_div(n_local, 4), 1u << 20), 256>>>(
            sh->offsets.p, idx->offsets.p, k, n_local, dim, idx->view(), idx->codes.p, idx->factors.p, idx->map_ids.p,
            sh->view(), sh->codes.p, sh->factors.p, sh->map_ids.p);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    RQC(finish_index(sh.get()));
    *out = sh.release();
    return RQ_OK;
}

// RCCL is bound at first use, not at link time: the library loads (and every single-GPU entry works) on a host
// without RCCL, and a host that already carries RCCL (a Rust binary linked against it, torch) shares its copy, so
// the communicator handle and the collective come from the same library.  rq_set_collectives replaces the three
// functions by the host's own (same signatures): another transport, or a test harness.
struct RcclApi {
    int (*all_gather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*user_rank)(void *, int *) = nullptr;
    const char *(*error_string)(int) = nullptr;
    std::string err;
};
static RcclApi g_custom_coll;
static std::atomic<bool> g_use_custom_coll{false};
static RcclApi *rccl_api() {
    if (g_use_custom_coll.load()) return &g_custom_coll;
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        const char *env = getenv("RABITQ_RCCL_LIB");
        if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        void *sym = h ? dlsym(h, "ncclAllGather") : dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (!sym && !h) {
            for (const char *name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
            sym = h ? dlsym(h, "ncclAllGather") : nullptr;
        }
        if (!sym) {
            api.err = "RCCL not found (ncclAllGather): set RABITQ_RCCL_LIB to the librccl.so the communicator came from";
            return;
        }
        auto find = [&](const char *name) { return h ? dlsym(h, name) : dlsym(RTLD_DEFAULT, name); };
        api.all_gather = reinterpret_cast<decltype(api.all_gather)>(sym);
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(find("ncclAllReduce"));
        api.user_rank = reinterpret_cast<decltype(api.user_rank)>(find("ncclCommUserRank"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(find("ncclGetErrorString"));
    });
    return &api;
}
#define RQ_NCCL_INT32 2    // ncclInt32 (rccl.h: ncclDataType_t)
#define RQ_NCCL_UINT64 5   // ncclUint64
#define RQ_NCCL_FLOAT32 7  // ncclFloat32
#define RQ_NCCL_MAX 2      // ncclMax (rccl.h: ncclRedOp_t)
#define RQ_NCCL_MIN 3      // ncclMin

static void profile_add(rq_profile_t &acc, const rq_profile_t &x) {
    acc.ms_rotate += x.ms_rotate, acc.ms_coarse += x.ms_coarse, acc.ms_select += x.ms_select, acc.ms_prep += x.ms_prep;
    acc.ms_group += x.ms_group, acc.ms_scan += x.ms_scan, acc.ms_rerank += x.ms_rerank, acc.ms_sort += x.ms_sort;
    acc.ms_replay += x.ms_replay, acc.ms_total += x.ms_total, acc.scan_bytes += x.scan_bytes;
    acc.scan_candidates += x.scan_candidates, acc.rerank_candidates += x.rerank_candidates, acc.scan_launches += x.scan_launches;
    acc.retries += x.retries, acc.ms_scan_matrix += x.ms_scan_matrix, acc.matrix_launches += x.matrix_launches;
    acc.matrix_pairs += x.matrix_pairs, acc.matrix_subtile_steps += x.matrix_subtile_steps;
    acc.matrix_exact_steps += x.matrix_exact_steps, acc.rerank_shadow_rejects += x.rerank_shadow_rejects;
    acc.ms_early += x.ms_early, acc.small_batch_passes += x.small_batch_passes;
    acc.survivor_workspace_bytes = std::max(acc.survivor_workspace_bytes, x.survivor_workspace_bytes), acc.segmented_passes += x.segmented_passes;
    acc.matrix_additive_launches += x.matrix_additive_launches;
    acc.coarse_fallback_rows += x.coarse_fallback_rows;
}

// probe lists <-> merge keys (f32 distance bits << 32 | list id: distances are >= 0, so the bits order like the values;
// the padding (0xFFFFFFFF, +inf) of rq_coarse_topk_device sorts last)
__global__ void pack_probe_keys_kernel(const uint32_t *__restrict__ pc, const float *__restrict__ pd, uint64_t cells,
                                       unsigned long long *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells) keys[i] = ((unsigned long long)__builtin_bit_cast(uint32_t, pd[i]) << 32) | pc[i];
}
// merged keys (nq x npb) -> the whole probe list, its nearest list alone, the rest (shared-threshold step)
__global__ void unpack_probe_keys_kernel(const unsigned long long *__restrict__ keys, uint32_t nq, uint32_t npb,
                                         uint32_t *__restrict__ pc, float *__restrict__ pd, uint32_t *__restrict__ pc_a,
                                         float *__restrict__ pd_a, uint32_t *__restrict__ pc_b, float *__restrict__ pd_b) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nq * npb) return;
    const uint32_t b = (uint32_t)(i / npb), c = (uint32_t)(i - (uint64_t)b * npb);
    const unsigned long long key = keys[i];
    const uint32_t id = (uint32_t)key;
    const float dv = __builtin_bit_cast(float, (uint32_t)(key >> 32));
    pc[i] = id, pd[i] = dv;
    if (!pc_a) return;
    if (c == 0) pc_a[b] = id, pd_a[b] = dv;
    else pc_b[(uint64_t)b * (npb - 1) + c - 1] = id, pd_b[(uint64_t)b * (npb - 1) + c - 1] = dv;
}
// status words of the ranks (the last key of every rank's block of the final all-gather): any non-zero one -> flag
__global__ void gather_status_kernel(const unsigned long long *__restrict__ gathered, uint32_t world, uint64_t rank_stride,
                                     uint64_t at, uint32_t *__restrict__ flag) {
    uint32_t bad = 0;
    for (uint32_t w = 0; w < world; ++w) bad |= gathered[(uint64_t)w * rank_stride + at] != 0ull ? (w + 1u) : 0u;
    *flag = bad;
}

// The multi-GPU step.  Every rank holds all (replicated) rotated centroids and a subset of the lists.
//   0  handshake: ONE ncclAllReduce(max) of three int32 {h, -h, error} with h = a hash of the call's parameters: a rank
//      that failed validation or allocation, or was called with other parameters, makes EVERY rank return an error
//      before any data collective is issued (mismatched counts would hang or corrupt the gather buffers);
//   1  coarse ranking, sliced by queries: rank r ranks queries [r nq / world, (r+1) nq / world) against ALL lists (the
//      centroids are replicated), ONE all-gather of nq / world x nprobe (distance, list) keys per rank: every rank holds the
//      global probe lists, each exactly the single-index ranking of its query;
//   2  with shared thresholds (a shard's own threshold is looser than the reference's; a shard that does not hold a
//      query's neighbourhood would re-rank most of what it scans):
//        A  the nearest list alone (only its owner finds candidates): the usual staged pass;
//           seed = the k-th best distance of A where A is full, one ncclAllReduce(min) of nq floats;
//        B  the other probed lists, seeded (one stage);
//      else the whole probe list in one probed pass;
//   3  ONE all-gather of the per-shard top-k keys (+ one status word per rank), k-way merge on every rank.
// After the handshake a rank that fails locally KEEPS taking part in every collective (contributing f32::MAX thresholds
// and empty keys) and reports through its status word, so no peer is left blocked in a collective and all ranks return
// an error for the step.
static rq_status sharded_step(rq_index *mi, void *nccl_comm, uint32_t world, uint32_t id_offset, const float *d_queries,
                              uint32_t nq, uint32_t len, uint32_t probe, uint32_t topk, bool heuristic, float *d_out_dist,
                              uint32_t *d_out_id, uint32_t *d_out_n, bool shared) {
    RcclApi *api = nccl_comm ? rccl_api() : nullptr;
    if (api && (!api->all_gather || !api->all_reduce))
        return fail(RQ_ERR_UNSUPPORTED, api->err.empty() ? "ncclAllGather / ncclAllReduce not found" : api->err);
    if (api && world > 1 && !api->user_rank) return fail(RQ_ERR_UNSUPPORTED, "ncclCommUserRank not found");
    auto nccl_fail = [&](const char *what, int rc) {
        return fail(RQ_ERR_HIP, std::string(what) + ": " + (api && api->error_string ? api->error_string(rc) : "error " + std::to_string(rc)));
    };
    const uint32_t npb = std::min(probe, std::max(mi->k, 1u));
    const uint32_t width = shared ? 2 * topk : topk;  // keys per query and rank in the final all-gather
    Workspace *ws = ws_acquire(mi);
    struct Rel {
        rq_index *i;
        Workspace *w;
        ~Rel() { ws_release(i, w); }
    } rel{mi, ws};
    if (!ws->stream) HIPC(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));
    hipStream_t st = ws->stream;
    RQC(ws->sh_flag.ensure(8));  // [0..3] handshake, [4] status of the final gather
    // ---- 0. local validation + every buffer of the step, then the handshake ---------------------------------------------
    const uint64_t cells = (uint64_t)nq * topk, pcells = (uint64_t)nq * npb;
    const uint64_t out_stride = (uint64_t)nq * width + 1;  // a rank's block of the final all-gather: keys + status word
    int my_rank = 0;
    rq_status err = validate_query(mi, d_queries, len, probe, topk, d_out_dist, d_out_id, d_out_n);
    const bool sliced = api && world > 1;
    const uint64_t pchunk = (uint64_t)((nq + world - 1) / world) * npb;  // probe-list keys a rank contributes (query-sliced coarse ranking)
    auto alloc_all = [&]() -> rq_status {
        RQC(ws->sh_dist.ensure(cells));
        RQC(ws->sh_id.ensure(cells));
        RQC(ws->sh_n.ensure(nq));
        RQC(ws->sh_packed.ensure(std::max<uint64_t>(out_stride, sliced ? pchunk : 0)));
        RQC(ws->sh_gathered.ensure(std::max<uint64_t>(out_stride, sliced ? pchunk : 0) * world));
        RQC(ws->sh_merged.ensure(cells));
        RQC(ws->sh_pc.ensure(2 * pcells + nq));
        RQC(ws->sh_pd.ensure(2 * pcells + nq));
        if (shared) {
            RQC(ws->sh_dist_b.ensure(cells));
            RQC(ws->sh_id_b.ensure(cells));
            RQC(ws->sh_n_b.ensure(nq));
            RQC(ws->sh_thr.ensure(nq));
        }
        return RQ_OK;
    };
    if (err == RQ_OK) err = alloc_all();
    std::string err_msg = err != RQ_OK ? g_err : std::string();
    if (api) {
        if (world > 1 && api->user_rank) {
            const int rc = api->user_rank(nccl_comm, &my_rank);
            if (rc != 0 || my_rank < 0 || (uint32_t)my_rank >= world) {
                if (err == RQ_OK) err = RQ_ERR_INVALID, err_msg = "ncclCommUserRank failed or rank >= world";
                my_rank = 0;
            }
        }
        uint32_t h = 0x9E3779B9u;
        for (uint32_t v : {nq, len, probe, topk, world, (uint32_t)heuristic, (uint32_t)shared, mi->k, mi->dim}) h = (h ^ v) * 0x01000193u;
        const int32_t hs = (int32_t)(h & 0x3FFFFFFFu);
        const int32_t hand[4] = {hs, -hs, err != RQ_OK ? 1 : 0, 0};
        // a local HIP failure here must not keep this rank out of the collective (its peers would block in it): it is folded into
        // `err`, the all-reduce is issued regardless (the words then on the device make the peers' parameter check fail), and
        // this rank returns its own error afterwards
        {
            const hipError_t he = hipMemcpyAsync(ws->sh_flag.p, hand, sizeof hand, hipMemcpyHostToDevice, st);
            if (he != hipSuccess && err == RQ_OK) err = RQ_ERR_HIP, err_msg = std::string("hipMemcpyAsync (handshake): ") + hipGetErrorString(he);
        }
        const int rc = api->all_reduce(ws->sh_flag.p, ws->sh_flag.p, 4, RQ_NCCL_INT32, RQ_NCCL_MAX, nccl_comm, st);
        if (rc != 0) return nccl_fail("ncclAllReduce (handshake)", rc);
        int32_t got[4] = {0, 0, 1, 0};
        {
            hipError_t he = hipMemcpyAsync(got, ws->sh_flag.p, sizeof got, hipMemcpyDeviceToHost, st);
            if (he == hipSuccess) he = hipStreamSynchronize(st);
            if (he != hipSuccess && err == RQ_OK) err = RQ_ERR_HIP, err_msg = std::string("handshake read-back: ") + hipGetErrorString(he);
        }
        if (err != RQ_OK) return fail(err, err_msg);
        if (got[0] != -got[1])
            return fail(RQ_ERR_INVALID, "rq_query_batch_sharded_device: the ranks were called with different nq / len / probe / topk / world / ranker / shared_thresholds");
        if (got[2] != 0) return fail(RQ_ERR_INVALID, "rq_query_batch_sharded_device: another rank failed before the step (its rq_last_error has the reason)");
    } else if (err != RQ_OK) {
        return fail(err, err_msg);
    }
    // from here on: no early return between collectives; a local failure is carried in `err`
    auto note = [&](rq_status s) {
        if (s != RQ_OK && s != RQ_ERR_EMPTY && err == RQ_OK) err = s, err_msg = g_err;
    };
    auto note_hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && err == RQ_OK) err = RQ_ERR_HIP, err_msg = std::string(what) + ": " + hipGetErrorString(e);
    };
    rq_profile_t prof_sum;
    memset(&prof_sum, 0, sizeof prof_sum);
    uint32_t *pc = ws->sh_pc.p, *pc_a = pc + pcells, *pc_b = pc_a + nq;
    float *pd = ws->sh_pd.p, *pd_a = pd + pcells, *pd_b = pd_a + nq;
    // ---- 1. probe lists ----------------------------------------------------------------------------------------------------
    const bool need_lists = sliced || shared;  // the plain unsliced step ranks inside query_device
    if (need_lists) {
        if (sliced) {
            // The coarse ranking is sliced by QUERIES: rank r ranks queries [r * chunk, (r + 1) * chunk) against ALL k lists, one
            // all-gather hands every rank every query's probe list.  (Until round 3 the slices were LISTS -- every rank ranked all
            // queries against its k / world lists, all-gathered world x nprobe keys per query and merged them: with the batch
            // growing with the world that is world x the keys on the wire -- 268 MB per rank at 8 x 65 536 queries -- plus a
            // 512-key merge per query; by queries it is 33 MB per rank and no merge, the same distance flops, and the ranking of a
            // query is literally the single-index ranking.)
            const uint32_t chunk = (nq + world - 1) / world, q_lo = std::min<uint64_t>((uint64_t)my_rank * chunk, nq);
            const uint32_t q_n = std::min<uint32_t>(chunk, nq - q_lo);
            note_hip(hipMemsetAsync(ws->sh_packed.p, 0xFF, (uint64_t)chunk * npb * 8, st), "hipMemsetAsync");  // an empty or failed slice: no list
            if (err == RQ_OK && q_n) {
                note_hip(hipStreamSynchronize(st), "hipStreamSynchronize");
                note(rq_coarse_topk_device(mi, d_queries + (uint64_t)q_lo * len, q_n, len, 0, mi->k, npb, pc, pd));  // synchronous, on a pooled workspace
                if (err == RQ_OK) pack_probe_keys_kernel<<<ceil_div((uint64_t)q_n * npb, 256), 256, 0, st>>>(pc, pd, (uint64_t)q_n * npb, ws->sh_packed.p);
            }
            const int rc = api->all_gather(ws->sh_packed.p, ws->sh_gathered.p, (uint64_t)chunk * npb, RQ_NCCL_UINT64, nccl_comm, st);
            if (rc != 0) note(nccl_fail("ncclAllGather (probe lists)", rc));
            // the gathered blocks are the probe lists of queries 0 .. world * chunk in order (rows past nq are padding)
            unpack_probe_keys_kernel<<<ceil_div(pcells, 256), 256, 0, st>>>(ws->sh_gathered.p, nq, npb, pc, pd, shared ? pc_a : nullptr, pd_a, pc_b, pd_b);
        } else if (err == RQ_OK) {
            note(rq_coarse_topk_device(mi, d_queries, nq, len, 0, mi->k, npb, pc, pd));
            if (shared) split_probe_kernel<<<ceil_div(pcells, 256), 256, 0, st>>>(pc, pd, nq, npb, pc_a, pd_a, pc_b, pd_b);
        }
        note_hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
    // ---- 2. this shard's answer --------------------------------------------------------------------------------------------
    note_hip(hipMemsetAsync(ws->sh_packed.p, 0xFF, (out_stride - 1) * 8, st), "hipMemsetAsync");  // nothing found (yet)
    if (shared) {
        if (err == RQ_OK) {
            note(query_device(mi, d_queries, nq, len, 1, topk, heuristic, ws->sh_dist.p, ws->sh_id.p, ws->sh_n.p, pc_a, pd_a, ws));
            profile_add(prof_sum, g_profile);
        }
        if (err == RQ_OK) kth_threshold_kernel<<<ceil_div(nq, 256), 256, 0, st>>>(ws->sh_dist.p, ws->sh_n.p, nq, topk, ws->sh_thr.p);
        else fill_f32_kernel<<<ceil_div(nq, 256), 256, 0, st>>>(ws->sh_thr.p, 3.402823466e+38f, nq);
        if (api) {
            const int rc = api->all_reduce(ws->sh_thr.p, ws->sh_thr.p, nq, RQ_NCCL_FLOAT32, RQ_NCCL_MIN, nccl_comm, st);
            if (rc != 0) note(nccl_fail("ncclAllReduce (thresholds)", rc));
        }
        note_hip(hipStreamSynchronize(st), "hipStreamSynchronize");
        if (err == RQ_OK) pack_topk_keys_at_kernel<<<ceil_div(cells, 256), 256, 0, st>>>(ws->sh_dist.p, ws->sh_id.p, ws->sh_n.p, nq, topk, id_offset, width, 0, ws->sh_packed.p);
        if (err == RQ_OK && npb > 1) {
            note(query_device(mi, d_queries, nq, len, npb - 1, topk, heuristic, ws->sh_dist_b.p, ws->sh_id_b.p, ws->sh_n_b.p, pc_b, pd_b, ws, ws->sh_thr.p));
            profile_add(prof_sum, g_profile);
            if (err == RQ_OK) pack_topk_keys_at_kernel<<<ceil_div(cells, 256), 256, 0, st>>>(ws->sh_dist_b.p, ws->sh_id_b.p, ws->sh_n_b.p, nq, topk, id_offset, width, topk, ws->sh_packed.p);
        }
    } else if (err == RQ_OK) {
        // every rank walks the same probe list; a list another rank owns is simply empty here (skipped before any per-pair work)
        note(query_device(mi, d_queries, nq, len, probe, topk, heuristic, ws->sh_dist.p, ws->sh_id.p, ws->sh_n.p, need_lists ? pc : nullptr,
                          need_lists ? pd : nullptr, ws));
        profile_add(prof_sum, g_profile);
        if (err == RQ_OK) pack_topk_keys_kernel<<<ceil_div(cells, 256), 256, 0, st>>>(ws->sh_dist.p, ws->sh_id.p, ws->sh_n.p, nq, topk, id_offset, ws->sh_packed.p);
    }
    // ---- 3. one all-gather of nq x width u64 keys + a status word per rank (latency-bound on xGMI), k-way merge -------------
    const unsigned long long status_word = err != RQ_OK ? 1ull : 0ull;
    note_hip(hipMemcpyAsync(ws->sh_packed.p + (out_stride - 1), &status_word, 8, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
    const unsigned long long *gathered = ws->sh_packed.p;
    if (api) {  // also with a communicator of one rank (the collective then copies)
        const int rc = api->all_gather(ws->sh_packed.p, ws->sh_gathered.p, out_stride, RQ_NCCL_UINT64, nccl_comm, st);
        if (rc != 0) note(nccl_fail("ncclAllGather (top-k)", rc));
        gathered = ws->sh_gathered.p;
    }
    const uint32_t gw = api ? world : 1u;
    merge_smallest_u64_kernel<<<nq, 256, (size_t)pow2_ceil(gw * width) * 8, st>>>(gathered, gw, nq, width, topk, ws->sh_merged.p, out_stride);
    unpack_topk_keys_kernel<<<ceil_div(nq, 256), 256, 0, st>>>(ws->sh_merged.p, nq, topk, d_out_dist, d_out_id, d_out_n);
    gather_status_kernel<<<1, 1, 0, st>>>(gathered, gw, out_stride, out_stride - 1, ws->sh_flag.p + 4);
    uint32_t bad = 0;
    note_hip(hipMemcpyAsync(&bad, ws->sh_flag.p + 4, 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
    note_hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    note_hip(hipGetLastError(), "kernel launch");
    g_profile = prof_sum;
    if (err != RQ_OK) return fail(err, err_msg);
    if (bad) return fail(RQ_ERR_HIP, "rq_query_batch_sharded_device: rank " + std::to_string(bad - 1) + " failed during the step (its rq_last_error has the reason)");
    return RQ_OK;
}

rq_status rq_query_batch_sharded_device(const rq_index *shard, void *nccl_comm, uint32_t world, uint32_t id_offset,
                                        const float *d_queries, uint32_t nq, uint32_t len, uint32_t probe, uint32_t topk,
                                        int heuristic_rank, float *d_out_dist, uint32_t *d_out_id, uint32_t *d_out_n) {
    RQC(ensure_device());
    RQC(ensure_kernel_attributes());
    if (!shard || world == 0) return fail(RQ_ERR_INVALID, "null argument");
    if (world > 1 && !nccl_comm) return fail(RQ_ERR_INVALID, "world > 1 needs an RCCL communicator");
    // limits that depend on the arguments only: every rank takes the same branch here
    if (topk == 0 || topk > RQ_MAX_TOPK || (uint64_t)world * topk > 8192) return fail(RQ_ERR_UNSUPPORTED, "topk in [1, 2048] and world * topk <= 8192");
    if ((uint64_t)nq * topk >= (1ull << 32) || (uint64_t)nq * std::min(probe, RQ_MAX_PROBE) >= (1ull << 32))
        return fail(RQ_ERR_UNSUPPORTED, "nq * topk and nq * probe must stay below 2^32 per call: split the batch");
    if (nq == 0) return RQ_OK;
    rq_index *mi = const_cast<rq_index *>(shard);
    const int shared_opt = g_shared_thr.load();  // 0 never, 1 when there are other shards (default), 2 always (tests: one-rank communicator)
    const bool shared = nccl_comm && (shared_opt == 2 || (shared_opt == 1 && world > 1)) && std::min(probe, shard->k) > 1;
    return sharded_step(mi, nccl_comm, world, id_offset, d_queries, nq, len, probe, topk, heuristic_rank != 0, d_out_dist, d_out_id,
                        d_out_n, shared);
}

rq_status rq_set_collectives(const rq_collectives_t *c) {
    if (!c) {
        g_use_custom_coll = false;
        return RQ_OK;
    }
    if (c->struct_size < sizeof(rq_collectives_t) || !c->all_gather || !c->all_reduce || !c->comm_user_rank)
        return fail(RQ_ERR_INVALID, "rq_set_collectives: struct_size too small or a null function");
    g_use_custom_coll = false;
    g_custom_coll.all_gather = reinterpret_cast<decltype(g_custom_coll.all_gather)>(c->all_gather);
    g_custom_coll.all_reduce = reinterpret_cast<decltype(g_custom_coll.all_reduce)>(c->all_reduce);
    g_custom_coll.user_rank = c->comm_user_rank;
    g_custom_coll.error_string = nullptr;
    g_use_custom_coll = true;
    return RQ_OK;
}

