// host_state.h -- part of the host side of librabitq_hip.so (one translation unit: rabitq_hip.hip includes the host_*.h files in order;
// they are not stand-alone headers).  Errors, device buffers, metrics / profiling, the per-call workspace, struct rq_index, the process-global options and the kernel launch helpers.
#pragma once
// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static rq_status fail(rq_status s, const std::string &msg) {
    g_err = msg;
    return s;
}
#define HIPC(expr)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return fail(_e == hipErrorOutOfMemory ? RQ_ERR_OOM : RQ_ERR_HIP,                         \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                          \
    } while (0)
#define RQC(expr)                        \
    do {                                 \
        rq_status _s = (expr);           \
        if (_s != RQ_OK) return _s;      \
    } while (0)

static rq_status ensure_device() {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(RQ_ERR_NO_DEVICE, "no HIP device visible (librabitq_hip has no CPU fallback)");
    return RQ_OK;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
    rq_status alloc(size_t n) {
        release();
        count = n;
        if (n == 0) n = 1;
        hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            count = 0;
            return fail(RQ_ERR_OOM, "hipMalloc of " + std::to_string(n * sizeof(T)) + " bytes failed: " +
                                        hipGetErrorString(e));
        }
        return RQ_OK;
    }
    rq_status ensure(size_t n) { return n <= count && p ? RQ_OK : alloc(n); }
};

// Sized out-structs (include/rabitq_hip.h): write at most the bytes the caller's struct has.
template <typename T>
static rq_status copy_out_sized(T *out, T full) {
    const uint32_t sz = out->struct_size;
    if (sz < 8) return fail(RQ_ERR_INVALID, "struct_size is not set (set it to sizeof(the struct) before the call)");
    const uint32_t w = std::min<uint32_t>(sz, (uint32_t)sizeof(T));
    full.struct_size = w;
    memcpy(out, &full, w);
    return RQ_OK;
}
static inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
template <int W, int NT>
static size_t assign_lds_bytes() { return 2 * (32 * (64 * W * 2 + 16) + 128); }  // assign_approx_kernel: two centroid-tile images
static inline uint32_t pow2_ceil(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ------------------------------------------------------------------------------------------------
// metrics (src/metrics.rs:65: process-global relaxed atomics)
// ------------------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_rough{0}, g_precise{0}, g_query{0}, g_miss{0};

// ------------------------------------------------------------------------------------------------
// profiling
// ------------------------------------------------------------------------------------------------
enum { PF_ROTATE = 0, PF_COARSE, PF_SELECT, PF_PREP, PF_GROUP, PF_SCAN, PF_SCAN_MATRIX, PF_RERANK, PF_SORT, PF_REPLAY, PF_EARLY, PF_TOTAL, PF_N };
static std::atomic<int> g_profiling{0};
static thread_local rq_profile_t g_profile;

struct Prof {
    bool on = false;
    bool light = false;  // level 2: only the scan launches and the whole pass are bracketed
    hipStream_t stream = nullptr;
    struct Span {
        hipEvent_t a, b;
        int cat;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    bool open = false;          // the last begin() was recorded (not filtered out)
    hipEvent_t last_b = nullptr;  // end event of the previous span, reusable as the next begin while nothing ran since
    bool failed = false;  // an event could not be created: this pass reports no timings (never a wrong one)
    hipEvent_t get() {
        if (used == pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess || !e) {
                failed = true;
                on = false;
                return nullptr;
            }
            pool.push_back(e);
        }
        return pool[used++];
    }
    // An event record costs ~5 us of stream time: adjacent spans share their boundary event.
    void begin(int cat) {
        open = on && !(light && cat != PF_SCAN && cat != PF_SCAN_MATRIX && cat != PF_TOTAL);
        if (!open) {
            last_b = nullptr;
            return;
        }
        Span s{last_b, get(), cat};
        if (s.b && !s.a) {
            s.a = get();
            if (s.a) (void)hipEventRecord(s.a, stream);
        }
        last_b = nullptr;
        if (!s.a || !s.b) {  // event creation failed: profiling is off for the rest of the pass
            open = false;
            spans.clear();
            return;
        }
        spans.push_back(s);
    }
    void end() {
        if (!open) return;
        (void)hipEventRecord(spans.back().b, stream);
        last_b = spans.back().b;
        open = false;
    }
    void reset(int level, hipStream_t st) {
        on = level != 0;
        light = level == 2;
        stream = st;
        spans.clear();
        used = 0;
        open = false;
        last_b = nullptr;
        failed = false;
    }
    void collect(float *ms /*PF_N*/) {
        if (failed) return;
        for (auto &s : spans) {
            float t = 0;
            (void)hipEventElapsedTime(&t, s.a, s.b);
            ms[s.cat] += t;
        }
    }
    ~Prof() {
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

// ------------------------------------------------------------------------------------------------
// query workspace
// ------------------------------------------------------------------------------------------------
struct StreamRange {
    uint32_t s_lo, s_hi;
};
struct Workspace {
    hipStream_t stream = nullptr;
    bool busy = false;
    // context of a pass that has been enqueued but not yet finished (finish_pass)
    size_t pend_total_span = 0;
    uint64_t pend_seg_slots = 0;  // slots of the final stage's segments (0: uniform geometry)
    uint32_t pend_cap = 0;        // uniform capacity of the pass
    uint32_t pend_nq = 0;
    std::vector<StreamRange> pend_matrix_ranges;  // stream ranges scanned on the matrix cores (profiling only)
    DevBuf<float> qpad, y, dist, probe_dist, thr, recent;
    DevBuf<float> retry_q, retry_pd, retry_pc;  // overflow re-runs: the affected queries (and their probe lists)
    DevBuf<uint32_t> retry_rows;
    DevBuf<uint32_t> q_hist, q_start, q_order;  // rerank order of a large batch (queries grouped by nearest list)
    DevBuf<uint32_t> live_list;                 // sharded passes: the (query, list) pairs whose list has members here, + their count
    DevBuf<uint32_t> coarse_redo;               // pre-filtered coarse ranking over more than 8192 lists: rows left to the block-per-query selection
    DevBuf<uint32_t> pair_rank, rank_base;      // group_rank_kernel: places of a big stage's pairs inside their groups
    DevBuf<uint32_t> probe_cluster, recs, grp_cnt, grp_start, heap_len, heap_id, precise, need,
        nsurv, nshadow, win_count, arr_len, row_map, big_list;
    DevBuf<int32_t> heap_key;
    DevBuf<PairScalars> scal;
    DevBuf<uint64_t> planes;
    DevBuf<uint32_t> qnib;
    DevBuf<uint32_t> qf6;
    DevBuf<unsigned long long> rough_cnt, totals, surv_cnt, stat;
    DevBuf<float4> grp_vref;  // additive gate: per list, centre and half-range of v' over the stage's pairs (group_vrange_kernel)
    bool pend_additive = false;  // the pass ran a matrix-core stage with the additive gate (finish_pass reads its flag rate)
    uint32_t pend_matrix_stages = 0;  // matrix-core stages of the pass
    bool pend_prefiltered = false;    // the pass ranked its lists through the matrix-core pre-filter (totals[12] = rows that fell back)
    DevBuf<SurvRec> surv, arr;
    DevBuf<RunRec> runs, runs_tmp;
    bool use_runs_tmp = false;
    bool arena_failed = false;  // the last pass gave up inside an arena stage (no room for the arena): the caller repeats it on the uniform buffers
    // multi-GPU step (rq_query_batch_sharded_device): this shard's results, the all-gathered keys, the merged keys
    DevBuf<float> sh_dist;
    DevBuf<uint32_t> sh_id, sh_n;
    DevBuf<unsigned long long> sh_packed, sh_gathered, sh_merged;
    DevBuf<uint32_t> ovf, q_cap;               // per query: overflow flag; segment capacity of the final stage (segmented passes)
    DevBuf<unsigned long long> q_base;         // per query: first slot of its segment
    DevBuf<SurvRec> arena_recs;                // arena stages: survivors of all queries, unordered (256 shards)
    DevBuf<RunRec> arena_runs;                 //   their run descriptors as uint4 {pos, slot | cnt << 16, query, offset}
    DevBuf<uint2> arena_places;                //   per descriptor: the run's place in its query's segment {first record, directory slot}
    DevBuf<unsigned long long> arena_cur;      //   RQ_ARENA_SHARDS shard cursors (records | runs << 32), overflow flag, total, cursor of the common area
    DevBuf<unsigned int> arena_fail;           //   per shard: first run index it turned away
    DevBuf<ScanExtra> scan_extra;              //   what the scan reads on its survivor path in arena mode
    DevBuf<uint32_t> sh_flag;                 // handshake / status words of the step
    DevBuf<uint32_t> sh_pc, sh_id_b, sh_n_b;  // shared-threshold step: probe lists (whole | nearest | rest), second call's results
    DevBuf<float> sh_pd, sh_thr, sh_dist_b;
    unsigned long long *h_totals = nullptr;  // pinned, 16
    Prof prof;
    ~Workspace() {
        if (h_totals) (void)hipHostFree(h_totals);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

struct rq_index {
    uint32_t dim = 0, k = 0, W = 0, max_list_len = 0;
    uint32_t min_list_len = 0;  // 0 if some list is empty (then no slot bound can be derived from stream positions)
    uint64_t n = 0;
    // raw vectors (cluster order, un-rotated).  Untiered (n_dev == n, the usual case): row p at base + p*dim.  Tiered
    // (they do not fit the HBM budget): per list the first h_c members in HBM, the tail in pinned host memory mapped
    // into the device address space (BaseView / ListTier); n_dev = sum of h_c.
    uint64_t n_dev = 0;
    float *base_host = nullptr;      // hipHostMalloc'ed (mapped); host address
    float *base_host_dev = nullptr;  // the same memory as the kernels address it
    DevBuf<ListTier> list_tier;      // k entries, tiered indexes only
    std::vector<ListTier> h_list_tier;
    bool split_rows = false;         // the raw vectors are split rows (common.h; set with the tiers, tiered indexes only)
    BaseView view() const { return BaseView{base.p, base_host_dev, list_tier.p, k, split_rows ? 1u : 0u}; }
    ~rq_index() {
        if (base_host) (void)hipHostFree(base_host);
    }
    DevBuf<float> base, P, centroids, cent_t;
    DevBuf<_Float16> base_h;  // fp16 shadow of `base` (rerank pre-filter, derived; untiered indexes with HBM to spare; option rerank_shadow = 1)
    DevBuf<uint8_t> base_q8;  // 8-bit shadow of `base`, one affine map per list (the default pre-filter: half the fp16 shadow's bytes per survivor)
    DevBuf<float4> list_q8;   //   per list: {lo, s, max |x_i - x^_i| over the list's rows, -}
    DevBuf<uint32_t> offsets, map_ids;
    DevBuf<uint64_t> codes;
    DevBuf<float4> factors;
    DevBuf<float4> list_uref;  // per list: mean of u' = (1, cds, ., eb) / factor_ip over its regular vectors (additive gate of the matrix-core scan; derived)
    DevBuf<uint16_t> cent_bf;  // k x dim bf16 image of the rotated centroids and their squared norms (coarse pre-filter; derived)
    DevBuf<float> cent_sqnorm;
    float cent_norm_max = INFINITY;  // largest centroid norm (inf: no pre-filter)
    uint32_t nonempty_lists = 0;  // lists with at least one vector (a shard of a multi-GPU index owns only some of the k lists)
    std::atomic<int> additive_loose{0};  // the additive gate flagged too many sub-tile steps on this index: later passes use the bf16 threshold
    std::mutex ws_mu;
    std::vector<std::unique_ptr<Workspace>> ws_pool;
    FactorStats fstats{0, 0, 0, 0};
    std::atomic<uint32_t> cap_hint{0};  // survivor-buffer capacity learnt from earlier batches
    std::atomic<uint64_t> arena_hint{0};  // slots the largest arena stage of earlier batches needed (+ headroom)
    std::atomic<uint32_t> big_dirs_hint{0};  // most long run directories (> 512 runs) a stage of a recent pass produced
    uint64_t pass_budget = 24ull << 30;  // bytes of survivor / run buffers one query pass may use (set by finish_index)
    // tile tables of the cluster-major scans: per tile size, one {list, first, list begin, list length} entry per
    // existing (list, tile); built on first use from the host copy of the offsets
    std::vector<uint32_t> h_offsets;
    std::mutex tt_mu;
    std::map<uint32_t, std::unique_ptr<DevBuf<uint4>>> tile_tables;
};

// ------------------------------------------------------------------------------------------------
// small init kernels
// ------------------------------------------------------------------------------------------------
__global__ void fill_f32_kernel(float *p, float v, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void gather_rows_kernel(const float *__restrict__ in, const uint32_t *__restrict__ rows,
                                   uint32_t nrows, uint32_t len, float *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nrows * len) return;
    uint32_t r = (uint32_t)(i / len), c = (uint32_t)(i - (uint64_t)r * len);
    out[i] = in[(uint64_t)rows[r] * len + c];
}

// ------------------------------------------------------------------------------------------------
// multi-GPU helpers: carving a shard out of an index, (distance, id) <-> u64 merge keys
// ------------------------------------------------------------------------------------------------
// one wave per destination position p of the shard: its list c is found by bisection over the shard's offsets,
// its source position is old_offsets[c] + (p - new_offsets[c])
__global__ __launch_bounds__(256) void shard_gather_kernel(const uint32_t *__restrict__ new_off, const uint32_t *__restrict__ old_off,
                                                           uint32_t k, uint64_t n_local, uint32_t dim,
                                                           const BaseView base_in, const uint64_t *__restrict__ codes_in,
                                                           const float4 *__restrict__ factors_in, const uint32_t *__restrict__ ids_in,
                                                           const BaseView base_out, uint64_t *__restrict__ codes_out,
                                                           float4 *__restrict__ factors_out, uint32_t *__restrict__ ids_out) {
    const uint32_t lane = threadIdx.x & 63, W = dim >> 6;
    for (uint64_t p = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < n_local; p += (uint64_t)gridDim.x * 4) {
        uint32_t lo = 0, hi = k;  // largest c with new_off[c] <= p (empty lists share their start with the next one)
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (new_off[mid] <= p) lo = mid;
            else hi = mid;
        }
        const uint64_t src = (uint64_t)old_off[lo] + (p - new_off[lo]);
        const RowRef srow = base_in.row(src, dim), drow = base_out.row(p, dim);
        for (uint32_t e = lane; e < dim; e += 64) rq_row_put(drow, dim, e, rq_row_get(srow, dim, e));
        for (uint32_t w = lane; w < W; w += 64) codes_out[p * W + w] = codes_in[src * W + w];
        if (lane == 0) {
            factors_out[p] = factors_in[src];
            ids_out[p] = ids_in[src];
        }
    }
}
// per-shard top-k -> merge keys (Ord32 image << 32 | global id); entries past the valid count sort last
__global__ void pack_topk_keys_kernel(const float *__restrict__ dist, const uint32_t *__restrict__ id, const uint32_t *__restrict__ cnt,
                                      uint32_t nq, uint32_t topk, uint32_t id_offset, unsigned long long *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nq * topk) return;
    const uint32_t b = (uint32_t)(i / topk), e = (uint32_t)(i - (uint64_t)b * topk);
    keys[i] = e < cnt[b] ? (((unsigned long long)ord32_biased(dist[i]) << 32) | (uint32_t)(id[i] + id_offset)) : ~0ull;
}
// shared-threshold multi-GPU step: the merged probe lists split into the nearest list and the rest
__global__ void split_probe_kernel(const uint32_t *__restrict__ pc, const float *__restrict__ pd, uint32_t nq, uint32_t npb,
                                   uint32_t *__restrict__ pc_a, float *__restrict__ pd_a, uint32_t *__restrict__ pc_b,
                                   float *__restrict__ pd_b) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nq * npb) return;
    const uint32_t b = (uint32_t)(i / npb), c = (uint32_t)(i - (uint64_t)b * npb);
    if (c == 0) pc_a[b] = pc[i], pd_a[b] = pd[i];
    else pc_b[(uint64_t)b * (npb - 1) + c - 1] = pc[i], pd_b[(uint64_t)b * (npb - 1) + c - 1] = pd[i];
}
// a query's seed threshold: the k-th best distance its nearest list gave, if the list gave k; f32::MAX otherwise
__global__ void kth_threshold_kernel(const float *__restrict__ dist, const uint32_t *__restrict__ cnt, uint32_t nq, uint32_t topk,
                                     float *__restrict__ thr) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nq) return;
    float m = 3.402823466e+38f;
    if (cnt[b] == topk) {
        m = dist[(uint64_t)b * topk];
        for (uint32_t e = 1; e < topk; ++e) m = dist[(uint64_t)b * topk + e] > m ? dist[(uint64_t)b * topk + e] : m;
    }
    thr[b] = m;
}
// per-shard top-k -> merge keys, written at columns [col0, col0 + topk) of rows of `width` keys
__global__ void pack_topk_keys_at_kernel(const float *__restrict__ dist, const uint32_t *__restrict__ id, const uint32_t *__restrict__ cnt,
                                         uint32_t nq, uint32_t topk, uint32_t id_offset, uint32_t width, uint32_t col0,
                                         unsigned long long *__restrict__ keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)nq * topk) return;
    const uint32_t b = (uint32_t)(i / topk), e = (uint32_t)(i - (uint64_t)b * topk);
    keys[(uint64_t)b * width + col0 + e] =
        e < cnt[b] ? (((unsigned long long)ord32_biased(dist[i]) << 32) | (uint32_t)(id[i] + id_offset)) : ~0ull;
}
__global__ void unpack_topk_keys_kernel(const unsigned long long *__restrict__ keys, uint32_t nq, uint32_t topk,
                                        float *__restrict__ dist, uint32_t *__restrict__ id, uint32_t *__restrict__ cnt) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nq) return;
    uint32_t c = 0;
    for (uint32_t e = 0; e < topk; ++e) {
        const unsigned long long key = keys[(uint64_t)b * topk + e];
        if (key == ~0ull) break;  // ascending: the padding comes last
        dist[(uint64_t)b * topk + e] = ord32_unbias((uint32_t)(key >> 32));
        id[(uint64_t)b * topk + e] = (uint32_t)key;
        ++c;
    }
    cnt[b] = c;
}

// coarse ranking distances (src/rabitq.rs:283-287), every query against the lists [first, first + k) of cent_t
static std::atomic<int> g_coarse_impl{0};  // 0 auto, 1 LDS-broadcast kernels, 2 scalar-register kernel, 3 bf16-MFMA pre-filter + exact refinement wherever it applies (row in registers up to 8192 lists), 4 the same with the tile-minima selection wherever it applies
static std::atomic<int> g_scan_dbg{0};
// the probe selection runs one wave per query (row in registers) for these shapes, one block per query otherwise
static bool select_is_wave(uint32_t k, uint32_t nprobe, uint32_t nq) { return nprobe <= 64 && k <= 8192 && nq >= 8; }
static void launch_coarse(const float *cent_t, const float *y, float *dist, uint32_t k, uint32_t dim, uint32_t nq,
                          uint32_t kstride, hipStream_t st) {
    const int impl = g_coarse_impl.load();
    if ((impl == 2 || (impl == 0 && nq >= 2048)) && nq > 0)  // many queries per list: query side in SGPRs
        coarse_dist_sreg_kernel<8><<<dim3(ceil_div(nq, 8), ceil_div(k, 256)), 256, 0, st>>>(cent_t, y, dist, k, dim, nq, kstride);
    else if (nq >= 64 && dim <= 2048)  // 8 queries per thread: 8 packed VALU ops per centroid element loaded
        coarse_dist_kernel<8><<<dim3(ceil_div(nq, 8), ceil_div(k, 256)), 256, 8 * dim * sizeof(float), st>>>(cent_t, y, dist, k, dim, nq,
                                                                                                    kstride);
    else
        coarse_dist_kernel<4><<<dim3(ceil_div(nq, 4), ceil_div(k, 256)), 256, 4 * dim * sizeof(float), st>>>(cent_t, y, dist, k, dim, nq,
                                                                                                    kstride);
}

// Coarse ranking of nq rotated queries against ALL k lists: the matrix-core pre-filter + exact-order refinement where it applies
// (coarse_impl 3, or -- once measured faster -- auto for big batches), else the exact-order distance kernels + selection.
static std::atomic<int> g_pair_split{1};  // sharded passes: pairs of empty lists settled by a thread each, quantisation over the listed others (0 = lane group per pair: test hook)
static std::atomic<int> g_coarse_tiled_from{4096};  // pre-filtered coarse ranking: list count from which the selection goes through tile minima (developer knob)
static bool coarse_prefilter_has(uint32_t W) { return W == 1 || W == 2 || W == 3 || W == 4 || W == 6 || W == 8 || W == 12; }
// (more lists than one wave holds in registers -- the ranking of a multi-GPU deployment is over the lists of ALL shards -- go through
// the tile-minima selection, select_refine_tiled_kernel)
static bool coarse_prefilter_applies(const rq_index *idx, uint32_t nq, uint32_t nprobe) {
    const int impl = g_coarse_impl.load();
    return (impl == 3 || impl == 4 || (impl == 0 && nq >= 2048)) && coarse_prefilter_has(idx->W) && std::isfinite(idx->cent_norm_max) &&
           idx->cent_bf.p != nullptr && nprobe <= 64 && nq >= 8 && idx->k >= 64 && nprobe >= 1 && idx->k <= 65536;
}
// redo: nq flags (only written / read when k > 8192)
// y_bf: room for nq x dim bf16 (the query rows pre-rounded for the wide instantiation; any workspace buffer that is free at this point)
static void launch_coarse_prefiltered(const rq_index *idx, const float *y, float *dist, uint32_t nq, uint32_t nprobe, uint32_t *out_cluster,
                                      float *out_dist, uint32_t out_stride, unsigned long long *fallback_rows, uint32_t *redo, hipStream_t st,
                                      uint16_t *y_bf) {
    const uint32_t k = idx->k, dim = idx->dim;
    if (idx->W > 8) to_bf16_kernel<<<ceil_div((uint64_t)nq * dim / 8, 256), 256, 0, st>>>(y, (uint64_t)nq * dim, y_bf);
#define RQ_CAP(WW, NT)                                                                                                      \
    coarse_approx_kernel<WW, NT><<<ceil_div(nq, 128 * NT), 256, assign_lds_bytes<WW, NT>(), st>>>(y, idx->cent_bf.p, idx->cent_sqnorm.p, nq, \
                                                                                                  k, dist, y_bf)
    switch (idx->W) {
        case 1: RQ_CAP(1, 2); break;
        case 2: RQ_CAP(2, 2); break;
        case 3: RQ_CAP(3, 1); break;
        case 4: RQ_CAP(4, 1); break;
        case 6: RQ_CAP(6, 1); break;
        case 8: RQ_CAP(8, 1); break;
        default: RQ_CAP(12, 1); break;
    }
#undef RQ_CAP
    const dim3 g(ceil_div(nq, 4)), b(256);
    // the selection: tile minima (select_refine_tiled_kernel) wherever a row has at least nprobe tiles of 32 lists -- measured faster than
    // the register-resident row from 4096 lists up, and the only form beyond 8192 --, else the row in registers
    const uint32_t ntile = ceil_div(k, 32u);
    const int impl = g_coarse_impl.load();
    // (dim 768 and beyond stay on the row in registers below 8192 lists: 2.45 against 2.77 ms per 32 768 queries on the 100M x 768 index --
    // the margin of the pre-filter grows with the dimension, so more tiles are read again.  Round 4 saw 22 ms here: the time sat between
    // the launches behind coarse_approx_kernel<12,1>, which then spilled 142 registers -- a dispatch that needs more scratch than the queue
    // holds is set up and torn down around the launch -- and needs no scratch any more.)
    const bool tiled = redo != nullptr && ntile >= nprobe &&
                       (k > 8192 || impl == 4 || (impl != 3 && k >= (uint32_t)g_coarse_tiled_from.load() && idx->W <= 8));
    if (tiled) {
#define RQ_TILED(TPL)                                                                                                             \
    select_refine_tiled_kernel<TPL><<<g, b, 4 * 64 * (TPL) * 4, st>>>(dist, y, idx->centroids.p, idx->cent_norm_max, k, dim, nprobe, out_cluster, \
                                                                    out_dist, out_stride, nq, redo, fallback_rows)
        if (ntile <= 128) RQ_TILED(2);
        else if (ntile <= 256) RQ_TILED(4);
        else if (ntile <= 512) RQ_TILED(8);
        else if (ntile <= 1024) RQ_TILED(16);
        else RQ_TILED(32);
#undef RQ_TILED
        // rows the tiled kernel could not handle hold exact-order distances now: the block-per-query selection takes them (it exits at once for the others)
        select_probe_kernel<<<nq, 256, (size_t)nprobe * 8, st>>>(dist, k, nprobe, out_cluster, out_dist, 0u, out_stride, redo);
    } else if (k <= 1024)
        select_refine_wave_kernel<16><<<g, b, 0, st>>>(dist, y, idx->centroids.p, idx->cent_norm_max, k, dim, nprobe, out_cluster, out_dist, out_stride, nq, fallback_rows);
    else if (k <= 4096)
        select_refine_wave_kernel<64><<<g, b, 0, st>>>(dist, y, idx->centroids.p, idx->cent_norm_max, k, dim, nprobe, out_cluster, out_dist, out_stride, nq, fallback_rows);
    else
        select_refine_wave_kernel<128><<<g, b, 0, st>>>(dist, y, idx->centroids.p, idx->cent_norm_max, k, dim, nprobe, out_cluster, out_dist, out_stride, nq, fallback_rows);
}

// ------------------------------------------------------------------------------------------------
// rotation launcher (MFMA kernel for bulk, VALU kernel for a handful of rows; bit-identical)
// ------------------------------------------------------------------------------------------------
// HIP silently wraps a launch whose gridDim.x * blockDim.x reaches 2^32: every launcher keeps
// blocks * threads below this bound (rows are chunked, big kernels are grid-stride).
#define RQ_MAX_BLOCKS_256 ((1u << 23) - 1)  // blocks of 256 threads: < 2^31 threads per launch

static void launch_rotate(const float *x, const float *P, float *out, uint64_t n, uint32_t dim, bool mfma,
                          hipStream_t st) {
    const uint64_t rows_per_launch = mfma ? (1ull << 40) : (uint64_t)RQ_MAX_BLOCKS_256 * 4;
    for (uint64_t r0 = 0; r0 < n; r0 += rows_per_launch) {
        const uint64_t m = std::min(rows_per_launch, n - r0);
        const float *xs = x + r0 * dim;
        float *os = out + r0 * dim;
        if (mfma) {
            // persistent: 2 blocks per CU x 256 CUs, split between the column tiles
            const uint32_t ncol = dim / ROT_BN;
            const uint64_t nrow_tiles = ceil_div(m, ROT_BM);
            const uint32_t groups = (uint32_t)std::min<uint64_t>(nrow_tiles, std::max<uint32_t>(1, 512 / ncol));
            rotate_mfma_kernel<<<dim3(groups * ncol), dim3(256), 0, st>>>(xs, P, os, m, dim, groups);
        } else {
            rotate_valu_kernel<<<dim3(ceil_div(m, 4), dim / 64), dim3(64, 4), 0, st>>>(xs, P, os, m, dim);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// scan dispatch on W = dim / 64
// ------------------------------------------------------------------------------------------------
// The tile table of `tile` positions per block (see ScanArgs::use_table); nullptr on failure (the caller then uses the
// plain grid).  Built once per (index, tile size).
static const uint4 *get_tile_table(const rq_index *cidx, uint32_t tile, uint32_t *count) {
    rq_index *idx = const_cast<rq_index *>(cidx);
    std::lock_guard<std::mutex> lk(idx->tt_mu);
    auto it = idx->tile_tables.find(tile);
    if (it == idx->tile_tables.end()) {
        std::vector<uint4> h;
        h.reserve(idx->n / tile + idx->k + 1);
        for (uint32_t c = 0; c < idx->k; ++c) {
            const uint32_t b = idx->h_offsets[c], len = idx->h_offsets[c + 1] - b;
            for (uint32_t f = 0; f < len; f += tile) h.push_back(make_uint4(c, f, b, len));
        }
        std::unique_ptr<DevBuf<uint4>> buf(new DevBuf<uint4>());
        if (buf->alloc(h.size()) != RQ_OK) return nullptr;
        if (!h.empty() && hipMemcpy(buf->p, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        buf->count = h.size();
        it = idx->tile_tables.emplace(tile, std::move(buf)).first;
    }
    *count = (uint32_t)it->second->count;
    return it->second->p;
}

// A stage's grid is ngroups x tiles_per_group blocks.  Shapes whose grid exceeds the launch bound (few huge lists
// x many pairs, ~1e9 vectors with skewed lists) are issued as several launches over (group, tile) sub-ranges; the
// survivors of one stage are unordered until the run directory is sorted, so the split changes nothing.
static std::atomic<uint32_t> g_max_scan_blocks{RQ_MAX_BLOCKS_256};  // lowered by tests ("max_scan_blocks")
template <typename F>
static void launch_scan_chunks(const ScanArgs &a, F &&launch) {
    if (a.ngroups == 0 || a.tiles_per_group == 0) return;
    const uint32_t maxb = std::max(1u, g_max_scan_blocks.load());
    const uint32_t tchunk = std::min(a.tiles_per_group, maxb);
    const uint32_t gchunk = std::max(1u, maxb / tchunk);
    for (uint32_t t0 = 0; t0 < a.tiles_per_group; t0 += tchunk)
        for (uint32_t g0 = 0; g0 < a.ngroups; g0 += gchunk) {
            ScanArgs c = a;
            c.tile_base = t0, c.group_base = g0;
            c.tiles_per_group = std::min(tchunk, a.tiles_per_group - t0);
            c.ngroups = std::min(gchunk, a.ngroups - g0);
            launch(c, dim3(c.ngroups * c.tiles_per_group));
        }
}

#define SCAN_ARGS p.codes, p.factors, p.offsets, p.grp_start, p.recs, p.surv, p.runs, p.surv_cnt, p.tile_table, a
template <bool ARENA>
static void launch_scan_t(const ScanPtrs &p, const ScanArgs &args, uint32_t W, hipStream_t st) {
    launch_scan_chunks(args, [&](const ScanArgs &a, dim3 g) {
        const dim3 b(256);
        switch (W) {
            case 1: scan_kernel<1, 2, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            case 2: scan_kernel<2, 2, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            case 3: scan_kernel<3, 2, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            case 4: scan_kernel<4, 2, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            case 6: scan_kernel<6, 2, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            case 8: scan_kernel<8, 2, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            case 12: scan_kernel<12, 1, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            case 16: scan_kernel<16, 1, ARENA><<<g, b, 0, st>>>(SCAN_ARGS); break;
            default:
                if constexpr (!ARENA) scan_generic_kernel<<<g, b, 0, st>>>(SCAN_ARGS, W);  // (arena stages exist for the fused dims only)
                break;
        }
    });
}
// args.x != nullptr: the arena instantiations (the stage appends to the shared arena, ScanExtra)
static void launch_scan(const ScanPtrs &p, const ScanArgs &args, uint32_t W, hipStream_t st) {
    if (args.x) launch_scan_t<true>(p, args, W, st);
    else launch_scan_t<false>(p, args, W, st);
}
// scan implementation: 0 = auto (matrix cores when many queries share each list, VALU otherwise),
// 1 = VALU (v_dot8_u32_u4) only, 2 = matrix cores wherever the kernel exists (test hook)
static std::atomic<int> g_scan_impl{0};
// gate of the matrix-core scan: 0 = auto (additive bound where it exists -- dim 64 / 128, uniform survivor buffers -- unless the index has
// shown that it flags too much), 1 = the bf16 rank-5 threshold MFMA always, 2 = additive wherever it exists (test hook)
static std::atomic<int> g_scan_gate{0};
static std::atomic<int> g_stage_growth{0};  // 0 = default schedule
static std::atomic<int> g_large_from{256};  // queries from which a batch runs the large-batch form of the stages (full-chip rerank / ordering / replay launches, thin early stages, dense run directories, survivor arena); below: one fused launch per stage
static bool rq_large_batch(uint32_t nq) { return nq >= (uint32_t)g_large_from.load(); }
static std::atomic<int> g_cluster_major_div{32};  // a VALU stage goes list-major once its (query, list) pairs reach k / this
static std::atomic<int> g_stage_settle_pct{100};  // developer knob: where a large batch's early (VALU) stages end and the final (matrix-core) stage begins, in percent of the average list length
static std::atomic<int> g_scan_tile_table{1};  // 0 = plain (list x tile) grids everywhere (test / measurement hook)
static std::atomic<int> g_group_rank{1};  // group_rank_kernel for cluster-major stages: 0 never, 1 big stages, 2 always
static std::atomic<int> g_shared_thr{1};  // rq_query_batch_sharded_device: thresholds shared between the shards (0 never, 1 world > 1, 2 always)
static std::atomic<int> g_sb_span{2560};  // developer knob: stream positions a query's block scans itself at most (small-batch path)
static std::atomic<int> g_seg_opt{1};  // per-query survivor segments in the final stage: 0 never, 1 once the index has shown that the default capacity overflows, 2 every large batch (tests)
static std::atomic<int> g_pass_overlap{1};  // a call of several passes keeps two of them in flight (1, default) or runs them one after the other (0)
static std::atomic<int> g_small_batch{0};  // small-batch path (kernels_small.h): 0 = whenever it applies (default), 1 = never (test hook)
static std::atomic<int> g_dense_dir{1};  // dense run directories for the VALU stages of large batches (0 = always append + sort: test hook)

// matrix-core scan instantiations: W = dim/64, NT = 32-candidate sub-tiles per wave (resident operand registers
// 6*W*NT), blocks per CU per scan_mfma_blocks_per_cu<W>()
static bool scan_has_mfma(uint32_t W) {
    switch (W) {
        case 1: case 2: case 3: case 4: case 6: case 8: case 12: case 16: return true;
        default: return false;
    }
}
static uint32_t scan_mfma_nt(uint32_t W, bool additive = false) { return W == 2 ? (additive ? RQ_ADD_NT2 : RQ_NT_W2) : (W == 12 ? RQ_NT_W12 : (W >= 4 ? 2 : 4)); }
static uint32_t scan_mfma_nw(uint32_t W, bool arena) { return W == 2 && !arena ? 8u : 4u; }  // scan_mfma_waves<W, ARENA>()
static uint32_t scan_mfma_tile(uint32_t W, bool arena, bool additive = false) { return 32 * scan_mfma_nw(W, arena) * scan_mfma_nt(W, additive && !arena); }
static size_t scan_mfma_ring_bytes(uint32_t W, bool arena = false) {  // scan_mfma_ring_slots<W, ARENA>() tile images
    (void)arena;
    const uint64_t slots = W <= 2 ? 4ull : (W >= 16 ? 5ull : 3ull);
    return slots * (32 * (12 * W + 2) + RQ_REC_TAIL * 32) * 4;
}
template <int W, int NT, bool ARENA, bool ADD = false>
static void launch_scan_mfma_t(const ScanPtrs &p, const ScanArgs &a, dim3 g, hipStream_t st) {
    scan_mfma_kernel<W, NT, ARENA, ADD><<<g, dim3(64 * scan_mfma_waves<W, ARENA>()), scan_mfma_ring_bytes(W, ARENA), st>>>(p.codes, p.factors, p.offsets, p.grp_start, p.grp_cnt,
                                                                                    p.recs, p.surv, p.runs, p.surv_cnt, p.stat, p.tile_table, p.list_uref, p.grp_vref, a);
}
// the additive-gate instantiations (dim 64 / 128, uniform survivor buffers)
static bool scan_has_additive(uint32_t W) { return W == 1 || W == 2; }
static void launch_scan_mfma_add(const ScanPtrs &p, const ScanArgs &args, uint32_t W, hipStream_t st) {
    launch_scan_chunks(args, [&](const ScanArgs &a, dim3 g) {
        switch (W) {
            case 1: launch_scan_mfma_t<1, 4, false, true>(p, a, g, st); break;
            case 2: launch_scan_mfma_t<2, RQ_ADD_NT2, false, true>(p, a, g, st); break;
            default: break;
        }
    });
}
// callers check scan_has_mfma(W) first; args.x != nullptr: the arena instantiations
template <bool ARENA>
static void launch_scan_mfma_a(const ScanPtrs &p, const ScanArgs &args, uint32_t W, hipStream_t st) {
    launch_scan_chunks(args, [&](const ScanArgs &a, dim3 g) {
        switch (W) {
            case 1: launch_scan_mfma_t<1, 4, ARENA>(p, a, g, st); break;
            case 2: launch_scan_mfma_t<2, RQ_NT_W2, ARENA>(p, a, g, st); break;
            case 3: launch_scan_mfma_t<3, 4, ARENA>(p, a, g, st); break;
            case 4: launch_scan_mfma_t<4, 2, ARENA>(p, a, g, st); break;
            case 6: launch_scan_mfma_t<6, 2, ARENA>(p, a, g, st); break;
            case 8: launch_scan_mfma_t<8, 2, ARENA>(p, a, g, st); break;
            case 12: launch_scan_mfma_t<12, RQ_NT_W12, ARENA>(p, a, g, st); break;
            case 16: launch_scan_mfma_t<16, 2, ARENA>(p, a, g, st); break;
            default: break;
        }
    });
}
static void launch_scan_mfma(const ScanPtrs &p, const ScanArgs &args, uint32_t W, hipStream_t st, bool additive = false) {
    if (args.x) launch_scan_mfma_a<true>(p, args, W, st);
    else if (additive) launch_scan_mfma_add(p, args, W, st);
    else launch_scan_mfma_a<false>(p, args, W, st);
}

static bool scan_is_fused(uint32_t W) {
    switch (W) {
        case 1: case 2: case 3: case 4: case 6: case 8: case 12: case 16: return true;
        default: return false;
    }
}
static uint32_t scan_tile(uint32_t W) {
    switch (W) {
        case 1: case 2: case 3: case 4: case 6: case 8: return 512;
        default: return 256;
    }
}

// probe selection: one wave per query when the row fits in registers and nprobe <= 64, else one block per query
static void launch_select(const float *dist, uint32_t k, uint32_t nprobe, uint32_t *out_cluster, float *out_dist,
                          uint32_t id_offset, uint32_t out_stride, uint32_t nq, hipStream_t st) {
    if (select_is_wave(k, nprobe, nq)) {
        const dim3 g(ceil_div(nq, 4)), b(256);
        if (k <= 1024) select_probe_wave_kernel<16><<<g, b, 0, st>>>(dist, k, nprobe, out_cluster, out_dist, id_offset, out_stride, nq);
        else if (k <= 4096) select_probe_wave_kernel<64><<<g, b, 0, st>>>(dist, k, nprobe, out_cluster, out_dist, id_offset, out_stride, nq);
        else select_probe_wave_kernel<128><<<g, b, 0, st>>>(dist, k, nprobe, out_cluster, out_dist, id_offset, out_stride, nq);
        return;
    }
    select_probe_kernel<<<nq, 256, (size_t)nprobe * 8, st>>>(dist, k, nprobe, out_cluster, out_dist, id_offset, out_stride);
}

// ------------------------------------------------------------------------------------------------
// Kernels whose dynamic LDS can exceed the 64 KiB default: the attribute is set once per process, before the
// first launch of any of them (every entry point that can reach such a launch calls this first), and a refusal
// is reported instead of surfacing later as a failed launch.
// ------------------------------------------------------------------------------------------------
template <int W, int NT>
static hipError_t set_scan_mfma_attr() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan_mfma_kernel<W, NT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_mfma_ring_bytes(W));
    if (e != hipSuccess) return e;
    if constexpr (W <= 2) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan_mfma_kernel<W, (W == 2 ? RQ_ADD_NT2 : NT), false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_mfma_ring_bytes(W));
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(reinterpret_cast<const void *>(scan_mfma_kernel<W, NT, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_mfma_ring_bytes(W));
}
static rq_status ensure_kernel_attributes() {
    static std::once_flag once;
    static hipError_t err = hipSuccess;
    static const char *what = "";
    std::call_once(once, [] {
        auto set = [&](const void *fn, int bytes, const char *name) {
            if (err != hipSuccess) return;
            err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (err != hipSuccess) what = name;
        };
        set(reinterpret_cast<const void *>(select_probe_kernel), 140 * 1024, "select_probe_kernel");
        set(reinterpret_cast<const void *>(coarse_dist_kernel<4>), 140 * 1024, "coarse_dist_kernel<4>");
        set(reinterpret_cast<const void *>(coarse_dist_kernel<8>), 140 * 1024, "coarse_dist_kernel<8>");
        set(reinterpret_cast<const void *>(assign_generic_kernel<8>), 140 * 1024, "assign_generic_kernel<8>");
        set(reinterpret_cast<const void *>(merge_smallest_u64_kernel), 16384 * 8, "merge_smallest_u64_kernel");
        set(reinterpret_cast<const void *>(group_rank_kernel), 32768 * 4, "group_rank_kernel");
        set(reinterpret_cast<const void *>(sb_front_kernel), 140 * 1024, "sb_front_kernel");
        set(reinterpret_cast<const void *>(sort_runs_mid_kernel), RQ_SORT_MID_LDS_WORDS * 8, "sort_runs_mid_kernel");  // (+ 34 KiB of static LDS)
        set(reinterpret_cast<const void *>(assign_approx_kernel<6, 1>), (int)assign_lds_bytes<6, 1>(), "assign_approx_kernel<6,1>");
        set(reinterpret_cast<const void *>(assign_approx_kernel<8, 1>), (int)assign_lds_bytes<8, 1>(), "assign_approx_kernel<8,1>");
        set(reinterpret_cast<const void *>(assign_approx_kernel<12, 1>), (int)assign_lds_bytes<12, 1>(), "assign_approx_kernel<12,1>");
        set(reinterpret_cast<const void *>(coarse_approx_kernel<6, 1>), (int)assign_lds_bytes<6, 1>(), "coarse_approx_kernel<6,1>");
        set(reinterpret_cast<const void *>(coarse_approx_kernel<8, 1>), (int)assign_lds_bytes<8, 1>(), "coarse_approx_kernel<8,1>");
        set(reinterpret_cast<const void *>(coarse_approx_kernel<12, 1>), (int)assign_lds_bytes<12, 1>(), "coarse_approx_kernel<12,1>");
        set(reinterpret_cast<const void *>(sb_finish_kernel<true>), 104 * 1024, "sb_finish_kernel");   // (+ 33 KiB of static LDS)
        set(reinterpret_cast<const void *>(sb_finish_kernel<false>), 104 * 1024, "sb_finish_kernel");  // (+ 49 KiB of static LDS)
#define RQ_SBQ_ATTR(WW)                                                                                  \
    set(reinterpret_cast<const void *>(sb_query_kernel<WW, 0>), 120 * 1024, "sb_query_kernel");          \
    set(reinterpret_cast<const void *>(sb_query_kernel<WW, 1>), 120 * 1024, "sb_query_kernel");          \
    set(reinterpret_cast<const void *>(sb_query_kernel<WW, 2>), 120 * 1024, "sb_query_kernel")
        RQ_SBQ_ATTR(1);
        RQ_SBQ_ATTR(2);
        RQ_SBQ_ATTR(4);
        RQ_SBQ_ATTR(8);
        RQ_SBQ_ATTR(12);
        RQ_SBQ_ATTR(16);
#undef RQ_SBQ_ATTR
        auto chk = [&](hipError_t e, const char *name) {
            if (err == hipSuccess && e != hipSuccess) err = e, what = name;
        };
        chk(set_scan_mfma_attr<1, 4>(), "scan_mfma_kernel<1,4>");
        chk(set_scan_mfma_attr<2, RQ_NT_W2>(), "scan_mfma_kernel<2,NT>");
        chk(set_scan_mfma_attr<3, 4>(), "scan_mfma_kernel<3,4>");
        chk(set_scan_mfma_attr<4, 2>(), "scan_mfma_kernel<4,2>");
        chk(set_scan_mfma_attr<6, 2>(), "scan_mfma_kernel<6,2>");
        chk(set_scan_mfma_attr<8, 2>(), "scan_mfma_kernel<8,2>");
        chk(set_scan_mfma_attr<12, RQ_NT_W12>(), "scan_mfma_kernel<12,NT>");
        chk(set_scan_mfma_attr<16, 2>(), "scan_mfma_kernel<16,2>");
    });
    if (err != hipSuccess)
        return fail(RQ_ERR_HIP, std::string("hipFuncSetAttribute(") + what + "): " + hipGetErrorString(err));
    return RQ_OK;
}

