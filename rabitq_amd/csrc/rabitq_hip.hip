// rabitq_hip.hip -- host side of librabitq_hip.so: index state, build / query orchestration,
// persistence and the extern "C" boundary declared in include/rabitq_hip.h.
//
// There is NO CPU fallback anywhere in this file: every arithmetic step of the path runs in the
// gfx950 kernels of kernels_query.h / kernels_build.h, and every entry point fails with
// RQ_ERR_NO_DEVICE / RQ_ERR_HIP when no device is usable.
#include "../../include/rabitq_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <charconv>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <vector>
#include <sys/stat.h>
#include <dlfcn.h>
#include <queue>
#include <map>

#include "common.h"
#include "kernels_build.h"
#include "kernels_query.h"
#include "kernels_small.h"

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
    BaseView view() const { return BaseView{base.p, base_host_dev, list_tier.p, k}; }
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
        const float *srow = base_in.row(src, dim);
        float *drow = base_out.row_mut(p, dim);
        for (uint32_t e = lane; e < dim; e += 64) drow[e] = srow[e];
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
            if (lo < 2 * avg && hi > 2 * avg) hi = 2 * avg;
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

    // 3. per-pair query quantisation (:304-317)
    pf.begin(PF_PREP);
    {
        uint32_t *qn = scan_is_fused(W) ? ws.qnib.p : nullptr;
        uint32_t *q6 = scan_has_mfma(W) && impl != 1 ? ws.qf6.p : nullptr;
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
            HIPC(hipMemcpyAsync(&nlive, ws.live_list.p + npairs, 4, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
        }
#define RQ_PREP_SMALL(LP, R, PPB, PP)                                                                              \
    do {                                                                                                           \
        if (listed)                                                                                                \
            prep_small_listed_kernel<LP, R, PP><<<std::max(1u, ceil_div(nlive, (PPB) * (PP))), 256, 0, st>>>(ws.y.p, idx->centroids.p, idx->offsets.p, probe_cluster, \
                                                                    probe_dist, ws.live_list.p, nlive, nprobe, ws.scal.p, qn, q6, k); \
        else                                                                                                       \
            prep_small_kernel<LP, R, PP><<<ceil_div(npairs, (PPB) * (PP)), 256, 0, st>>>(ws.y.p, idx->centroids.p, idx->offsets.p, probe_cluster, \
                                                                    probe_dist, npairs, nprobe, ws.scal.p, qn, q6, k, idx->nonempty_lists * 2 < k ? 2u : 1u); \
    } while (0)
        if (dim == 128) RQ_PREP_SMALL(32, 1, 8, 4);  // 32 lanes per pair, two pairs per wave, four rounds of pairs per lane group
        else if (dim == 64) RQ_PREP_SMALL(16, 1, 16, 4);
        else if (dim == 256) RQ_PREP_SMALL(64, 1, 4, 4);
        else if (dim == 512) RQ_PREP_SMALL(64, 2, 4, 2);
        else if (dim == 768) RQ_PREP_SMALL(64, 3, 4, 2);
        else if (dim == 1024) RQ_PREP_SMALL(64, 4, 4, 2);
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

    // 5. stages
    if (one_stage) {
        stages.push_back({0u, 0xFFFFFFFFu});
    } else {
        // small batches are launch-bound (coarser stages), large ones rerank-bound (tighter thresholds)
        const int gopt = g_stage_growth.load();
        const uint64_t growth = gopt >= 2 ? (uint64_t)gopt : (rq_large_batch(nq) ? 8 : 16);
        // the first stage runs with threshold f32::MAX (everything survives) until the ranker's heap is full; in a large
        // batch it also takes what would be the next stage (whose threshold -- the worst of the first topk -- lets most
        // of it through anyway): one stage of launches less for ~1 % more exact distances
        stages = build_stages((uint64_t)std::max<uint32_t>(topk, 1) * (rq_large_batch(nq) ? growth : 1), growth, ~0ull);
    }
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
        const bool use_mfma = scan_has_mfma(W) && impl != 1 &&
                              (impl == 2 || (est_pairs >= 8ull * k && (sg.s_lo >= avg_len * (uint64_t)g_stage_settle_pct.load() / 100 || one_stage)));
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
            if (fill_items)
                stage_fill_kernel<<<ceil_div(fill_items, 16), 256, 0, st>>>(ws.scal.p, probe_cluster, operand, ws.thr.p, fill_items,
                                                                    nprobe, slot_hi, fp6_records ? 12 * W : 8 * W, sg.s_lo, sg.s_hi,
                                                                    a.cluster_major, ws.grp_start.p, ws.grp_cnt.p, ws.recs.p,
                                                                    idx->fstats, use_mfma ? (additive ? 2u : 1u) : 0u, ranked ? ws.pair_rank.p : nullptr,
                                                                    ws.rank_base.p, k, idx->list_uref.p, fill_listed ? ws.live_list.p : nullptr);
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
                    probe_cluster, nprobe, ws.nshadow.p);
            else if (idx->base_h.p && (stage_no > 0 || qp.thr_init) && !(g_scan_dbg & 512))
                accurate_filtered_kernel<<<dim3(gx, nq), 256, (size_t)dim * sizeof(float), st>>>(
                    ws.surv.p, ws.surv_cnt.p, seg, idx->base.p, idx->base_h.p, qpad, dim, rerank_order, ws.thr.p,
                    ws.nshadow.p);
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

// ------------------------------------------------------------------------------------------------
// index construction helpers
// ------------------------------------------------------------------------------------------------
// fp16 shadow of the raw vectors (kernels_query.h: rerank pre-filter).  Only for untiered indexes, and only when the
// 2*dim bytes per vector still leave the query workspaces their room; it costs one streaming pass over `base`.
#define RQ_SHADOW_MIN_ROWS 1ull
static std::atomic<int> g_rerank_shadow{2};  // shadow rows for the rerank pre-filter: 0 never, 1 fp16 when they fit, 2 8-bit (one affine map per list) when they fit
static rq_status derive_shadow_rows(rq_index *idx);
static rq_status finish_index(rq_index *idx) {
    // derived state: transposed centroids, longest list
    idx->W = idx->dim / 64;
    RQC(idx->cent_t.alloc((size_t)idx->dim * idx->k));
    RQC(ensure_kernel_attributes());
    if (idx->k)
        transpose_kernel<<<dim3(ceil_div(idx->dim, 32), ceil_div(idx->k, 32)), dim3(32, 8)>>>(
            idx->centroids.p, idx->cent_t.p, idx->k, idx->dim);
    DevBuf<uint32_t> mx;
    RQC(mx.alloc(2));
    const uint32_t mx_init[2] = {0u, 0xFFFFFFFFu};
    HIPC(hipMemcpy(mx.p, mx_init, 8, hipMemcpyHostToDevice));
    if (idx->k) max_list_len_kernel<<<ceil_div(idx->k, 256), 256>>>(idx->offsets.p, idx->k, mx.p);
    uint32_t mx_out[2];
    HIPC(hipMemcpy(mx_out, mx.p, 8, hipMemcpyDeviceToHost));
    idx->max_list_len = mx_out[0];
    idx->min_list_len = idx->k ? mx_out[1] : 0;
    {  // Factor bounds for the integer-threshold form of the gate
        DevBuf<uint32_t> st4;
        RQC(st4.alloc(4));
        HIPC(hipMemset(st4.p, 0, 16));
        if (idx->n)
            factor_stats_kernel<<<(uint32_t)std::min<uint64_t>(ceil_div(idx->n, 256), 4096), 256>>>(idx->factors.p, idx->n, st4.p);
        HIPC(hipMemcpy(&idx->fstats, st4.p, 16, hipMemcpyDeviceToHost));
    }
    idx->cent_norm_max = INFINITY;
    if (idx->k) {  // bf16 centroids + norms for the matrix-core pre-filter of the coarse ranking
        const uint64_t cells = (uint64_t)idx->k * idx->dim;
        DevBuf<uint32_t> mxb;
        RQC(mxb.alloc(1));
        HIPC(hipMemset(mxb.p, 0, 4));
        RQC(idx->cent_bf.alloc(cells));
        RQC(idx->cent_sqnorm.alloc(idx->k));
        to_bf16_kernel<<<ceil_div(cells, 2048), 256>>>(idx->centroids.p, cells, idx->cent_bf.p);
        row_sqnorm_kernel<<<ceil_div(idx->k, 256), 256>>>(idx->centroids.p, idx->k, idx->dim, idx->cent_sqnorm.p, mxb.p);
        float m2 = 0.0f;
        HIPC(hipMemcpy(&m2, mxb.p, 4, hipMemcpyDeviceToHost));
        idx->cent_norm_max = std::isfinite(m2) && m2 < 1.0e30f ? std::sqrt(m2) * 1.000001f : INFINITY;
    }
    if (idx->k) {  // per-list reference of the candidates' side of the additive gate
        RQC(idx->list_uref.alloc(idx->k));
        list_uref_kernel<<<idx->k, 256>>>(idx->factors.p, idx->offsets.p, idx->list_uref.p);
    }
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    RQC(derive_shadow_rows(idx));
    idx->h_offsets.resize((size_t)idx->k + 1);
    HIPC(hipMemcpy(idx->h_offsets.data(), idx->offsets.p, ((size_t)idx->k + 1) * 4, hipMemcpyDeviceToHost));
    idx->nonempty_lists = 0;
    for (uint32_t c = 0; c < idx->k; ++c) idx->nonempty_lists += idx->h_offsets[c + 1] > idx->h_offsets[c] ? 1u : 0u;
    {
        size_t free_b = 0, total_b = 0;
        HIPC(hipMemGetInfo(&free_b, &total_b));
        idx->pass_budget = std::min<uint64_t>(std::max<uint64_t>(free_b / 3, 4ull << 30), 96ull << 30);
    }
    return RQ_OK;
}

static rq_status derive_shadow_rows(rq_index *idx) {
    idx->base_h.release();
    idx->base_q8.release();
    idx->list_q8.release();
    int kind = g_rerank_shadow.load();
    if (kind == 2 && idx->dim > 4096) kind = 1;  // (the 8-bit encoder handles rows of up to 4096 dimensions: wider vectors take the fp16 rows)
    if (!kind || idx->base_host != nullptr || idx->n < RQ_SHADOW_MIN_ROWS) return RQ_OK;
    const uint64_t total = idx->n * idx->dim, bytes = total * (kind == 2 ? 1 : 2);
    size_t free_b = 0, total_b = 0;
    HIPC(hipMemGetInfo(&free_b, &total_b));
    if (free_b < bytes + (48ull << 30) && bytes > (1ull << 30)) return RQ_OK;  // keep the survivor buffers their share
    if (kind == 2) {  // one byte per dimension, per-list affine map, measured error bound (kernels_query.h)
        if (idx->base_q8.alloc(total) != RQ_OK || idx->list_q8.alloc(std::max<uint32_t>(idx->k, 1)) != RQ_OK) {
            (void)hipGetLastError();
            idx->base_q8.release();
            idx->list_q8.release();
            return RQ_OK;  // no room: queries run without the pre-filter
        }
        std::vector<uint32_t> off((size_t)idx->k + 1);
        HIPC(hipMemcpy(off.data(), idx->offsets.p, off.size() * 4, hipMemcpyDeviceToHost));
        uint32_t longest = 0;
        for (uint32_t c = 0; c < idx->k; ++c) longest = std::max(longest, off[c + 1] - off[c]);
        q8_range_kernel<<<idx->k, 256>>>(idx->base.p, idx->offsets.p, idx->dim, idx->list_q8.p);
        if (longest)
            for (uint32_t c0 = 0; c0 < idx->k; c0 += 32768)  // (grid.y is limited to 65535)
                q8_encode_kernel<<<dim3(ceil_div(longest, 256u), std::min(32768u, idx->k - c0)), 256>>>(
                    idx->base.p, idx->offsets.p + c0, idx->dim, idx->list_q8.p + c0, idx->base_q8.p);
        HIPC(hipDeviceSynchronize());
        HIPC(hipGetLastError());
        return RQ_OK;
    }
    if (idx->base_h.alloc(total) != RQ_OK) {
        (void)hipGetLastError();
        return RQ_OK;  // no room: queries run without the pre-filter
    }
    half_rows_kernel<<<(uint32_t)std::min<uint64_t>(ceil_div(total, 2048), 1u << 20), 256>>>(idx->base.p, total, idx->base_h.p);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    return RQ_OK;
}

// Gaussian-QR orthogonal matrix (src/utils.rs:16-20), seeded: Householder QR in f64 of a
// dim x dim N(0,1) matrix; Q returned row-major in f32.
static void gen_orthogonal(uint32_t dim, uint64_t seed, std::vector<float> &out) {
    std::mt19937_64 rng(seed);
    std::normal_distribution<double> nd(0.0, 1.0);
    const size_t D = dim;
    std::vector<double> A(D * D), Q(D * D, 0.0), v(D);
    for (auto &a : A) a = nd(rng);
    std::vector<std::vector<double>> vs;
    vs.reserve(D);
    for (size_t j = 0; j < D; ++j) {
        double norm = 0;
        for (size_t i = j; i < D; ++i) norm += A[i * D + j] * A[i * D + j];
        norm = std::sqrt(norm);
        std::vector<double> h(D, 0.0);
        double alpha = A[j * D + j] > 0 ? -norm : norm;
        for (size_t i = j; i < D; ++i) h[i] = A[i * D + j];
        h[j] -= alpha;
        double hn = 0;
        for (size_t i = j; i < D; ++i) hn += h[i] * h[i];
        if (hn > 0) {
            for (size_t c = j; c < D; ++c) {
                double dot = 0;
                for (size_t i = j; i < D; ++i) dot += h[i] * A[i * D + c];
                dot = 2 * dot / hn;
                for (size_t i = j; i < D; ++i) A[i * D + c] -= dot * h[i];
            }
        }
        vs.push_back(std::move(h));
    }
    for (size_t i = 0; i < D; ++i) Q[i * D + i] = 1.0;
    for (size_t jj = D; jj-- > 0;) {  // Q = H_0 H_1 ... H_{D-1}
        const auto &h = vs[jj];
        double hn = 0;
        for (size_t i = jj; i < D; ++i) hn += h[i] * h[i];
        if (hn == 0) continue;
        for (size_t c = 0; c < D; ++c) {
            double dot = 0;
            for (size_t i = jj; i < D; ++i) dot += h[i] * Q[i * D + c];
            dot = 2 * dot / hn;
            for (size_t i = jj; i < D; ++i) Q[i * D + c] -= dot * h[i];
        }
    }
    out.resize(D * D);
    for (size_t i = 0; i < D * D; ++i) out[i] = (float)Q[i];
}

static void launch_assign(const float *xrot, const rq_index *idx, uint64_t n, uint32_t *label, float *dist,
                          hipStream_t st) {
    if (n == 0) return;
    if (idx->dim == 128)
        assign_regs_kernel<128><<<ceil_div(n, 256), 256, 0, st>>>(xrot, idx->centroids.p, n, idx->k, label, dist);
    else if (idx->dim == 64)
        assign_regs_kernel<64><<<ceil_div(n, 256), 256, 0, st>>>(xrot, idx->centroids.p, n, idx->k, label, dist);
    else
        assign_generic_kernel<8><<<ceil_div(n, 8), 256, 8 * idx->dim * sizeof(float), st>>>(
            xrot, idx->cent_t.p, n, idx->k, idx->dim, label, dist);
}

// Nearest list through the matrix cores (kernels_build.h: assign_approx_kernel + assign_refine_kernel; exact results):
// what it needs besides the index's rotated centroids, built once per build, plus per-chunk scratch.
static std::atomic<int> g_assign_impl{0};  // 0 = matrix-core pre-filter where the kernel exists (default), 1 = exact-order VALU kernels only
struct AssignAux {
    DevBuf<uint16_t> cent_bf;  // k x dim bf16
    DevBuf<float> cnorm, redo_x, redo_dist;
    DevBuf<uint32_t> cand, cand_cnt, redo, redo_cnt, redo_lab;
    float cmax = INFINITY;
    uint64_t redone = 0;  // vectors that went through the exact-order kernel (no or too many candidates)
};
static bool assign_has_mfma(uint32_t W) { return W == 1 || W == 2 || W == 3 || W == 4 || W == 6 || W == 8 || W == 12; }
static rq_status assign_aux_init(const rq_index *idx, AssignAux &ax, uint64_t chunk_rows) {
    const uint64_t cells = (uint64_t)idx->k * idx->dim;
    RQC(ax.cent_bf.alloc(cells));
    RQC(ax.cnorm.alloc(idx->k));
    RQC(ax.redo_cnt.alloc(2));
    HIPC(hipMemset(ax.redo_cnt.p, 0, 8));
    to_bf16_kernel<<<ceil_div(cells, 2048), 256>>>(idx->centroids.p, cells, ax.cent_bf.p);
    row_sqnorm_kernel<<<ceil_div(idx->k, 256), 256>>>(idx->centroids.p, idx->k, idx->dim, ax.cnorm.p, ax.redo_cnt.p + 1);
    uint32_t bits = 0;
    HIPC(hipMemcpy(&bits, ax.redo_cnt.p + 1, 4, hipMemcpyDeviceToHost));
    const float m2 = __builtin_bit_cast(float, bits);
    ax.cmax = std::isfinite(m2) && m2 < 1.0e30f ? std::sqrt(m2) * 1.000001f : INFINITY;  // inf: every vector goes to the exact-order kernel
    RQC(ax.cand.alloc(chunk_rows * RQ_ASSIGN_CAND));
    RQC(ax.cand_cnt.alloc(chunk_rows));
    RQC(ax.redo.alloc(chunk_rows));
    return RQ_OK;
}
// n <= the chunk size given to assign_aux_init; null stream (the builder's)
static rq_status launch_assign_prefiltered(const float *xrot, const rq_index *idx, AssignAux &ax, uint64_t n, uint32_t *label,
                                           float *dist) {
    if (n == 0) return RQ_OK;
    const uint32_t W = idx->W, k = idx->k;
    if (g_assign_impl.load() == 1 || !assign_has_mfma(W) || !std::isfinite(ax.cmax)) {
        launch_assign(xrot, idx, n, label, dist, nullptr);
        return RQ_OK;
    }
    HIPC(hipMemsetAsync(ax.cand_cnt.p, 0, n * 4, nullptr));
    HIPC(hipMemsetAsync(ax.redo_cnt.p, 0, 4, nullptr));
#define RQ_ASG(WW, NT)                                                                                                   \
    assign_approx_kernel<WW, NT><<<ceil_div(n, 128 * NT), 256, assign_lds_bytes<WW, NT>(), nullptr>>>(                  \
        xrot, ax.cent_bf.p, ax.cnorm.p, ax.cmax, n, k, ax.cand.p, ax.cand_cnt.p)
    switch (W) {
        case 1: RQ_ASG(1, 2); break;
        case 2: RQ_ASG(2, 2); break;
        case 3: RQ_ASG(3, 1); break;
        case 4: RQ_ASG(4, 1); break;
        case 6: RQ_ASG(6, 1); break;
        case 8: RQ_ASG(8, 1); break;
        default: RQ_ASG(12, 1); break;
    }
#undef RQ_ASG
    assign_refine_kernel<<<ceil_div(2 * n, 256), 256>>>(xrot, idx->centroids.p, n, idx->dim, ax.cand.p, ax.cand_cnt.p, label, dist,
                                                        ax.redo.p, ax.redo_cnt.p);
    uint32_t m = 0;
    HIPC(hipMemcpy(&m, ax.redo_cnt.p, 4, hipMemcpyDeviceToHost));
    if (m) {  // no candidate (non-finite input) or more than RQ_ASSIGN_CAND of them: the exact-order kernel over all lists
        ax.redone += m;
        RQC(ax.redo_x.ensure((uint64_t)m * idx->dim));
        RQC(ax.redo_dist.ensure(m));
        RQC(ax.redo_lab.ensure(m));
        gather_rows_kernel<<<ceil_div((uint64_t)m * idx->dim, 256), 256>>>(xrot, ax.redo.p, m, idx->dim, ax.redo_x.p);
        launch_assign(ax.redo_x.p, idx, m, ax.redo_lab.p, ax.redo_dist.p, nullptr);
        assign_scatter_kernel<<<ceil_div(m, 256), 256>>>(ax.redo.p, m, ax.redo_lab.p, ax.redo_dist.p, label, dist);
    }
    return RQ_OK;
}

// ------------------------------------------------------------------------------------------------
// Base tiers: how many raw vectors stay in HBM.  budget_bytes: 0 = automatic (what is free now minus a reserve for
// query workspaces and the caller), ~0 = everything in HBM, else an explicit cap ("base_device_mb" option / the
// builder's argument).  The rest goes to pinned, device-mapped host memory.
// ------------------------------------------------------------------------------------------------
static std::atomic<int64_t> g_base_device_mb{-1};  // -1 = automatic
#define RQ_HBM_RESERVE_BYTES (12ull << 30)
// h_offsets: the k+1 list offsets on the host (the split is per list)
static rq_status alloc_base_tiers(rq_index *idx, uint64_t budget_bytes, const uint32_t *h_offsets) {
    const uint64_t row = (uint64_t)idx->dim * 4, want = idx->n * row;
    uint64_t cap = budget_bytes;
    if (budget_bytes == 0) {
        const int64_t opt = g_base_device_mb.load();
        if (opt >= 0) {
            cap = (uint64_t)opt << 20;
        } else {
            size_t free_b = 0, total_b = 0;
            HIPC(hipMemGetInfo(&free_b, &total_b));
            cap = free_b > RQ_HBM_RESERVE_BYTES ? free_b - RQ_HBM_RESERVE_BYTES : 0;
            if (want <= cap || want <= (256ull << 20)) cap = ~0ull;  // fits (or is small): no host tier
        }
    }
    const uint64_t budget_rows = cap == ~0ull ? idx->n : std::min<uint64_t>(idx->n, cap / row);
    if (budget_rows >= idx->n) {  // everything in HBM, rows at their positions
        idx->n_dev = idx->n;
        RQC(idx->base.alloc(idx->n * idx->dim));
        return RQ_OK;
    }
    // every list keeps the same share of its members (its head: the vectors nearest the centroid) in HBM
    const uint32_t k = idx->k;
    idx->h_list_tier.resize(k);
    uint64_t hbm = 0, host = 0;
    for (uint32_t c = 0; c < k; ++c) {
        const uint64_t len = h_offsets[c + 1] - h_offsets[c];
        const uint64_t h = idx->n ? len * budget_rows / idx->n : 0;  // floor: the sum never exceeds the budget
        idx->h_list_tier[c] = ListTier{h_offsets[c], (uint32_t)h, (uint32_t)hbm, (uint32_t)host};
        hbm += h, host += len - h;
    }
    idx->n_dev = hbm;
    RQC(idx->base.alloc(hbm * idx->dim));
    RQC(idx->list_tier.alloc(k));
    HIPC(hipMemcpy(idx->list_tier.p, idx->h_list_tier.data(), (size_t)k * sizeof(ListTier), hipMemcpyHostToDevice));
    hipError_t e = hipHostMalloc((void **)&idx->base_host, std::max<uint64_t>(host, 1) * row, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) {
        idx->base_host = nullptr;
        return fail(RQ_ERR_OOM, "pinned host tier of " + std::to_string(host * row) + " bytes: " + hipGetErrorString(e));
    }
    HIPC(hipHostGetDevicePointer((void **)&idx->base_host_dev, idx->base_host, 0));
    return RQ_OK;
}

// ------------------------------------------------------------------------------------------------
// Streamed two-pass build (RaBitQ::from_path, src/rabitq.rs:159-265, for inputs that need not be resident):
//   pass 1  rq_builder_assign_chunk  rotate (:188) -> nearest list (:203) -> sign-pack + factors (:205-229), per chunk
//           rq_builder_order         cluster ordering (:232-243), codes / factors / map_ids gathered, base tiers allocated
//   pass 2  rq_builder_place_chunk   raw vectors to their cluster-order positions (:244-247), HBM or host tier
//           rq_builder_finish        derived state, hand the index over
// The input is fed twice, chunk by chunk, so neither the n x d input nor a rotated copy ever has to coexist with the
// cluster-ordered base (the reference holds base + rotated copy + per-vector Vecs at once, :188-197).
// ------------------------------------------------------------------------------------------------
struct rq_builder {
    std::unique_ptr<rq_index> idx;
    uint32_t d = 0;
    uint64_t budget = 0;
    DevBuf<uint32_t> label, pos_of_id;
    DevBuf<float> mind, xpad, xrot;
    DevBuf<uint64_t> codes_tmp;
    DevBuf<float4> factors_tmp;
    AssignAux assign_aux;
    uint64_t assigned = 0, placed = 0;
    // Rows each pass has seen, as disjoint [begin, end) intervals: chunks may come in any order and size, but every row
    // exactly once per pass.  A duplicated chunk would leave other rows with uninitialised labels / codes (and then
    // index the list histogram with garbage), so overlap is refused here and gaps by the row counts in order / finish.
    struct Coverage {
        std::map<uint64_t, uint64_t> iv;  // begin -> end
        bool add(uint64_t i0, uint64_t m) {
            if (m == 0) return true;
            const uint64_t i1 = i0 + m;
            auto nx = iv.lower_bound(i0);  // first interval starting at or after i0
            if (nx != iv.end() && nx->first < i1) return false;
            if (nx != iv.begin()) {
                auto pv = std::prev(nx);
                if (pv->second > i0) return false;
                if (pv->second == i0) {  // extend the neighbour on the left (and swallow the one on the right if it touches)
                    pv->second = i1;
                    if (nx != iv.end() && nx->first == i1) pv->second = nx->second, iv.erase(nx);
                    return true;
                }
            }
            if (nx != iv.end() && nx->first == i1) {
                const uint64_t e = nx->second;
                iv.erase(nx);
                iv[i0] = e;
            } else {
                iv[i0] = i1;
            }
            return true;
        }
    } cov_assign, cov_place;
    bool ordered = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    rq_build_stats_t stats{};
    ~rq_builder() {
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};
#define RQ_BUILD_CHUNK (1ull << 20)

static rq_status builder_create(uint64_t n, uint32_t d, const float *d_centroids, uint32_t k, const float *orthogonal_host,
                                uint64_t seed, uint64_t max_device_base_bytes, rq_builder **out) {
    RQC(ensure_device());
    RQC(ensure_kernel_attributes());
    if (!out) return fail(RQ_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!d_centroids || d == 0 || k == 0) return fail(RQ_ERR_INVALID, "bad build arguments");
    if (n >= 0xFFFFFFFFull) return fail(RQ_ERR_UNSUPPORTED, "n must fit u32 ids (rabitq.rs:64-65)");
    const uint32_t dim = (d + 63) / 64 * 64;  // rabitq.rs:168-179
    if (dim > 4096) return fail(RQ_ERR_UNSUPPORTED, "dim > 4096 not supported");
    std::unique_ptr<rq_builder> b(new rq_builder());
    b->idx.reset(new rq_index());
    rq_index *idx = b->idx.get();
    idx->dim = dim, idx->k = k, idx->n = n, idx->W = dim / 64;
    b->d = d, b->budget = max_device_base_bytes;

    std::vector<float> Pgen;
    if (!orthogonal_host) {
        gen_orthogonal(dim, seed, Pgen);  // utils.rs:16-20, seeded
        orthogonal_host = Pgen.data();
    }
    RQC(idx->P.alloc((size_t)dim * dim));
    HIPC(hipMemcpy(idx->P.p, orthogonal_host, (size_t)dim * dim * 4, hipMemcpyHostToDevice));

    // centroids: pad, rotate (rabitq.rs:189), transpose for the lane<->centroid kernels
    DevBuf<float> cpad;
    RQC(cpad.alloc((size_t)k * dim));
    pad_rows_kernel<<<ceil_div((uint64_t)k * dim, 256), 256>>>(d_centroids, cpad.p, k, d, dim);
    RQC(idx->centroids.alloc((size_t)k * dim));
    launch_rotate(cpad.p, idx->P.p, idx->centroids.p, k, dim, true, nullptr);
    RQC(idx->cent_t.alloc((size_t)dim * k));
    transpose_kernel<<<dim3(ceil_div(dim, 32), ceil_div(k, 32)), dim3(32, 8)>>>(idx->centroids.p, idx->cent_t.p, k, dim);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());

    RQC(b->label.alloc(n));
    RQC(b->mind.alloc(n));
    RQC(b->codes_tmp.alloc(n * idx->W));
    RQC(b->factors_tmp.alloc(n));
    const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(n, 1), RQ_BUILD_CHUNK);
    if (d != dim) RQC(b->xpad.alloc(chunk * dim));
    RQC(b->xrot.alloc(chunk * dim));
    if (assign_has_mfma(idx->W) && g_assign_impl.load() != 1) RQC(assign_aux_init(idx, b->assign_aux, chunk));
    for (auto &e : b->ev) HIPC(hipEventCreate(&e));
    *out = b.release();
    return RQ_OK;
}

// pass 1 for rows [i0, i0 + m) of the input (d_rows: m x d, device)
static rq_status builder_assign(rq_builder *b, const float *d_rows, uint64_t i0, uint64_t m) {
    if (!b || (m && !d_rows)) return fail(RQ_ERR_INVALID, "null argument");
    if (b->ordered) return fail(RQ_ERR_INVALID, "rq_builder_assign_chunk after rq_builder_order");
    rq_index *idx = b->idx.get();
    if (i0 > idx->n || m > idx->n - i0) return fail(RQ_ERR_INVALID, "chunk outside [0, n)");
    if (!b->cov_assign.add(i0, m)) return fail(RQ_ERR_INVALID, "rq_builder_assign_chunk: rows [" + std::to_string(i0) + ", " + std::to_string(i0 + m) + ") overlap rows already assigned");
    const uint32_t d = b->d, dim = idx->dim;
    for (uint64_t c0 = 0; c0 < m; c0 += RQ_BUILD_CHUNK) {
        const uint64_t mm = std::min<uint64_t>(RQ_BUILD_CHUNK, m - c0), at = i0 + c0;
        const float *src = d_rows + c0 * d;
        if (d != dim) {
            pad_rows_kernel<<<ceil_div(mm * dim, 256), 256>>>(src, b->xpad.p, mm, d, dim);
            src = b->xpad.p;
        }
        HIPC(hipEventRecord(b->ev[0], nullptr));
        launch_rotate(src, idx->P.p, b->xrot.p, mm, dim, true, nullptr);
        HIPC(hipEventRecord(b->ev[1], nullptr));
        RQC(launch_assign_prefiltered(b->xrot.p, idx, b->assign_aux, mm, b->label.p + at, b->mind.p + at));
        HIPC(hipEventRecord(b->ev[2], nullptr));
        quantize_kernel<<<ceil_div(mm, 32), 256>>>(b->xrot.p, idx->centroids.p, b->label.p + at, mm, dim,
                                                   b->codes_tmp.p + at * idx->W, b->factors_tmp.p + at);
        HIPC(hipEventRecord(b->ev[3], nullptr));
        HIPC(hipEventSynchronize(b->ev[3]));  // the chunk buffers are reused by the next chunk (and by the caller)
        HIPC(hipGetLastError());
        float t01 = 0, t12 = 0, t23 = 0;
        HIPC(hipEventElapsedTime(&t01, b->ev[0], b->ev[1]));
        HIPC(hipEventElapsedTime(&t12, b->ev[1], b->ev[2]));
        HIPC(hipEventElapsedTime(&t23, b->ev[2], b->ev[3]));
        b->stats.ms_rotate += t01, b->stats.ms_assign += t12, b->stats.ms_quantize += t23;
    }
    b->assigned += m;
    b->stats.rows_assigned = b->assigned;
    b->stats.rows_exact_redo = b->assign_aux.redone;
    return RQ_OK;
}

// cluster ordering (rabitq.rs:232-252) of everything but the raw vectors
static rq_status builder_order(rq_builder *b) {
    if (!b) return fail(RQ_ERR_INVALID, "null builder");
    if (b->ordered) return fail(RQ_ERR_INVALID, "rq_builder_order called twice");
    rq_index *idx = b->idx.get();
    const uint64_t n = idx->n;
    const uint32_t k = idx->k;
    if (b->assigned != n) return fail(RQ_ERR_INVALID, "rq_builder_order before every row was assigned");
    b->xpad.release();
    b->xrot.release();
    {
        AssignAux &ax = b->assign_aux;
        ax.cent_bf.release(), ax.cnorm.release(), ax.redo_x.release(), ax.redo_dist.release(), ax.cand.release();
        ax.cand_cnt.release(), ax.redo.release(), ax.redo_cnt.release(), ax.redo_lab.release();
    }
    DevBuf<uint32_t> cnt;
    DevBuf<unsigned long long> keys;
    RQC(cnt.alloc((size_t)k + 1));
    RQC(idx->offsets.alloc((size_t)k + 1));
    RQC(keys.alloc(n));
    HIPC(hipMemset(cnt.p, 0, ((size_t)k + 1) * 4));
    const uint32_t g256 = (uint32_t)std::min<uint64_t>(ceil_div(std::max<uint64_t>(n, 1), 256), 1u << 22);
    if (n) label_hist_kernel<<<ceil_div(n, 256), 256>>>(b->label.p, n, cnt.p);
    group_scan_kernel<<<1, 1024>>>(cnt.p, k, idx->offsets.p, 0u, nullptr, 0u);  // also zeroes cnt -> cursor
    if (n) label_scatter_kernel<<<ceil_div(n, 256), 256>>>(b->label.p, b->mind.p, n, 0, idx->offsets.p, cnt.p, keys.p);
    list_sort_kernel<<<k, 1024>>>(keys.p, idx->offsets.p);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    b->label.release();
    b->mind.release();
    RQC(idx->codes.alloc(n * idx->W));
    RQC(idx->factors.alloc(n));
    RQC(idx->map_ids.alloc(n));
    RQC(b->pos_of_id.alloc(n));
    if (n)
        order_gather_kernel<<<g256, 256>>>(keys.p, n, idx->W, b->codes_tmp.p, b->factors_tmp.p, idx->codes.p, idx->factors.p,
                                           idx->map_ids.p, b->pos_of_id.p);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    b->codes_tmp.release();
    b->factors_tmp.release();
    keys.release();
    std::vector<uint32_t> h_off((size_t)k + 1);
    HIPC(hipMemcpy(h_off.data(), idx->offsets.p, ((size_t)k + 1) * 4, hipMemcpyDeviceToHost));
    RQC(alloc_base_tiers(idx, b->budget, h_off.data()));
    b->stats.rows_in_hbm = idx->n_dev, b->stats.rows_in_host_memory = n - idx->n_dev;
    b->ordered = true;
    return RQ_OK;
}

// pass 2 for rows [i0, i0 + m)
static rq_status builder_place(rq_builder *b, const float *d_rows, uint64_t i0, uint64_t m) {
    if (!b || (m && !d_rows)) return fail(RQ_ERR_INVALID, "null argument");
    if (!b->ordered) return fail(RQ_ERR_INVALID, "rq_builder_place_chunk before rq_builder_order");
    rq_index *idx = b->idx.get();
    if (i0 > idx->n || m > idx->n - i0) return fail(RQ_ERR_INVALID, "chunk outside [0, n)");
    if (!b->cov_place.add(i0, m)) return fail(RQ_ERR_INVALID, "rq_builder_place_chunk: rows [" + std::to_string(i0) + ", " + std::to_string(i0 + m) + ") overlap rows already placed");
    if (m) {
        place_rows_kernel<<<(uint32_t)std::min<uint64_t>(ceil_div(m, 4), 1u << 20), 256>>>(d_rows, i0, m, b->d, idx->dim,
                                                                                          b->pos_of_id.p, idx->view());
        HIPC(hipDeviceSynchronize());  // the caller may reuse d_rows right away
        HIPC(hipGetLastError());
    }
    b->placed += m;
    return RQ_OK;
}

static rq_status builder_finish(rq_builder *bp, rq_index **out) {
    if (!bp || !out) return fail(RQ_ERR_INVALID, "null argument");
    std::unique_ptr<rq_builder> b(bp);  // consumed whatever happens
    *out = nullptr;
    if (!b->ordered || b->placed != b->idx->n) return fail(RQ_ERR_INVALID, "rq_builder_finish before every row was placed");
    b->pos_of_id.release();
    RQC(finish_index(b->idx.get()));
    *out = b->idx.release();
    return RQ_OK;
}

static rq_status build_device(const float *d_base, uint64_t n, uint32_t d, const float *d_centroids, uint32_t k,
                              const float *orthogonal_host, uint64_t seed, rq_index **out) {
    if (!out) return fail(RQ_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (n && !d_base) return fail(RQ_ERR_INVALID, "bad build arguments");
    rq_builder *b = nullptr;
    RQC(builder_create(n, d, d_centroids, k, orthogonal_host, seed, 0, &b));
    std::unique_ptr<rq_builder> guard(b);
    RQC(builder_assign(b, d_base, 0, n));
    RQC(builder_order(b));
    RQC(builder_place(b, d_base, 0, n));
    return builder_finish(guard.release(), out);
}

// ------------------------------------------------------------------------------------------------
// "vecs" files (src/utils.rs:280-364): records [u32 LE count][count x elem LE]
// ------------------------------------------------------------------------------------------------
struct VecsFile {
    std::vector<unsigned char> data;  // concatenated payloads
    std::vector<uint32_t> lens;       // per-record element counts
};
static rq_status read_vecs_file(const std::string &path, size_t elem, VecsFile &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return fail(RQ_ERR_IO, "cannot open " + path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> raw((size_t)sz);
    if (sz && fread(raw.data(), 1, (size_t)sz, f) != (size_t)sz) {
        fclose(f);
        return fail(RQ_ERR_IO, "short read on " + path);
    }
    fclose(f);
    out.data.clear();
    out.lens.clear();
    out.data.reserve((size_t)sz);
    size_t off = 0;
    while (off + 4 <= (size_t)sz) {
        uint32_t cnt;
        memcpy(&cnt, raw.data() + off, 4);
        off += 4;
        size_t bytes = (size_t)cnt * elem;
        if (off + bytes > (size_t)sz) return fail(RQ_ERR_IO, "truncated record in " + path);
        out.data.insert(out.data.end(), raw.begin() + off, raw.begin() + off + bytes);
        out.lens.push_back(cnt);
        off += bytes;
    }
    return RQ_OK;
}
static rq_status write_record(FILE *f, const void *data, uint32_t count, size_t elem, const std::string &path) {
    if (fwrite(&count, 4, 1, f) != 1 || (count && fwrite(data, elem, count, f) != count))
        return fail(RQ_ERR_IO, "write error on " + path);
    return RQ_OK;
}

static rq_status copy_base_rows(const rq_index *idx, uint64_t i0, uint64_t m, float *buf, bool to_index);
static rq_status from_arrays(uint32_t dim, uint64_t n, uint32_t k, const float *base, const float *orthogonal,
                             const float *centroids, const uint32_t *offsets, const uint32_t *map_ids,
                             const uint64_t *codes, const rq_factor_t *factors, rq_index **out) {
    RQC(ensure_device());
    RQC(ensure_kernel_attributes());
    if (!out) return fail(RQ_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (dim == 0 || dim % 64 != 0) return fail(RQ_ERR_DIM_MISMATCH, "dim must be a non-zero multiple of 64 (rabitq.rs:109)");
    if (!orthogonal || !centroids || !offsets || (n && (!base || !map_ids || !codes || !factors)))
        return fail(RQ_ERR_INVALID, "null array");
    if (n >= 0xFFFFFFFFull) return fail(RQ_ERR_UNSUPPORTED, "n must fit u32");
    std::unique_ptr<rq_index> idx(new rq_index());
    idx->dim = dim, idx->n = n, idx->k = k, idx->W = dim / 64;
    RQC(idx->P.alloc((size_t)dim * dim));
    RQC(idx->centroids.alloc((size_t)k * dim));
    RQC(idx->offsets.alloc((size_t)k + 1));
    RQC(idx->map_ids.alloc(n));
    RQC(idx->codes.alloc(n * idx->W));
    RQC(idx->factors.alloc(n));
    for (uint32_t c = 0; c < k; ++c)
        if (offsets[c] > offsets[c + 1] || offsets[c + 1] > n) return fail(RQ_ERR_INVALID, "offsets are not a non-decreasing partition of [0, n]");
    if (k && offsets[k] != n) return fail(RQ_ERR_INVALID, "offsets[k] != n");
    RQC(alloc_base_tiers(idx.get(), 0, offsets));
    if (n) {
        RQC(copy_base_rows(idx.get(), 0, n, const_cast<float *>(base), /*to_index=*/true));
        HIPC(hipMemcpy(idx->map_ids.p, map_ids, n * 4, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(idx->codes.p, codes, n * idx->W * 8, hipMemcpyHostToDevice));
        HIPC(hipMemcpy(idx->factors.p, factors, n * 16, hipMemcpyHostToDevice));
    }
    HIPC(hipMemcpy(idx->P.p, orthogonal, (size_t)dim * dim * 4, hipMemcpyHostToDevice));
    if (k) HIPC(hipMemcpy(idx->centroids.p, centroids, (size_t)k * dim * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(idx->offsets.p, offsets, ((size_t)k + 1) * 4, hipMemcpyHostToDevice));
    RQC(finish_index(idx.get()));
    *out = idx.release();
    return RQ_OK;
}

// ---- JSON reader / writer of rq_load_json / rq_dump_json (serde_json image of `RaBitQ`, src/rabitq.rs:72-81) ----
namespace {
struct JsonOut {
    FILE *f;
    bool ok = true;
    void raw(const char *s) { ok = ok && fputs(s, f) >= 0; }
    void f32(float v) {
        if (!std::isfinite(v)) return raw("null");
        char buf[48];
        auto r = std::to_chars(buf, buf + 40, v);  // shortest representation that round-trips
        *r.ptr = 0;
        bool plain = true;
        for (char *c = buf; c < r.ptr; ++c) plain = plain && ((*c >= '0' && *c <= '9') || *c == '-');
        if (plain) strcpy(r.ptr, ".0");  // serde_json always marks a float ("1.0")
        raw(buf);
    }
    void u64(unsigned long long v) {
        char buf[32];
        snprintf(buf, sizeof buf, "%llu", v);
        raw(buf);
    }
    // Mat with nrows x ncols where element (i, j) = src[j * ld + i]  (col_major = our row-per-vector arrays) or src[i * ld + j]
    void mat(const float *src, uint64_t nrows, uint64_t ncols, bool transposed) {
        raw("{\"nrows\":"), u64(nrows), raw(",\"ncols\":"), u64(ncols), raw(",\"data\":[");
        for (uint64_t i = 0; i < nrows; ++i)
            for (uint64_t j = 0; j < ncols; ++j) {
                if (i || j) raw(",");
                f32(transposed ? src[j * nrows + i] : src[i * ncols + j]);
            }
        raw("]}");
    }
};
struct JsonIn {
    const char *p, *end;
    std::string err;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool lit(char c) {
        ws();
        if (p < end && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    bool need(char c) {
        if (lit(c)) return true;
        if (err.empty()) err = std::string("expected '") + c + "'";
        return false;
    }
    bool key(std::string &out) {
        ws();
        if (p >= end || *p != '"') return false;
        const char *q = ++p;
        while (p < end && *p != '"') ++p;
        if (p >= end) return false;
        out.assign(q, p);
        ++p;
        return need(':');
    }
    bool num_f32(float &v) {
        ws();
        if (end - p >= 4 && !strncmp(p, "null", 4)) {
            err = "null where a number is required (serde_json writes non-finite floats as null and cannot load them)";
            return false;
        }
        char *e = nullptr;
        v = strtof(p, &e);
        if (e == p) return false;
        p = e;
        return true;
    }
    bool num_u64(unsigned long long &v) {
        ws();
        char *e = nullptr;
        v = strtoull(p, &e, 10);
        if (e == p) return false;
        p = e;
        return true;
    }
    template <typename T, typename F>
    bool array(std::vector<T> &out, F &&one) {
        if (!need('[')) return false;
        if (lit(']')) return true;
        do {
            T v;
            if (!one(v)) return false;
            out.push_back(v);
        } while (lit(','));
        return need(']');
    }
    bool skip() {  // any value
        ws();
        if (p >= end) return false;
        if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            ++p;
            if (lit(close)) return true;
            do {
                if (open == '{') {
                    std::string k;
                    if (!key(k)) return false;
                }
                if (!skip()) return false;
            } while (lit(','));
            return need(close);
        }
        if (*p == '"') {
            ++p;
            while (p < end && *p != '"') p += (*p == '\\') ? 2 : 1;
            return p < end && *p++ == '"';
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']') ++p;
        return true;
    }
    bool mat(std::vector<float> &data, unsigned long long &nrows, unsigned long long &ncols) {
        if (!need('{')) return false;
        do {
            std::string k;
            if (!key(k)) return false;
            if (k == "nrows") {
                if (!num_u64(nrows)) return false;
            } else if (k == "ncols") {
                if (!num_u64(ncols)) return false;
            } else if (k == "data") {
                if (!array(data, [&](float &v) { return num_f32(v); })) return false;
            } else if (!skip()) {
                return false;
            }
        } while (lit(','));
        return need('}');
    }
};
}  // namespace

// rows [i0, i0 + m) of the cluster-ordered base between host memory (`buf`, m x dim) and whichever tier holds them
// (to_index = false: index -> buf; true: buf -> index)
static rq_status copy_base_rows(const rq_index *idx, uint64_t i0, uint64_t m, float *buf, bool to_index) {
    const uint64_t dim = idx->dim, i1 = i0 + m;
    auto dev_copy = [&](uint64_t dev_row, uint64_t rows, float *hp) -> rq_status {
        if (!rows) return RQ_OK;
        if (to_index) HIPC(hipMemcpy(idx->base.p + dev_row * dim, hp, rows * dim * 4, hipMemcpyHostToDevice));
        else HIPC(hipMemcpy(hp, idx->base.p + dev_row * dim, rows * dim * 4, hipMemcpyDeviceToHost));
        return RQ_OK;
    };
    if (!idx->base_host) return dev_copy(i0, m, buf);
    // tiered: walk the lists that overlap the range; per list an HBM piece and a host piece
    const std::vector<ListTier> &lt = idx->h_list_tier;
    uint32_t c = 0;
    {
        uint32_t lo = 0, hi = idx->k;
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (lt[mid].off <= i0) lo = mid;
            else hi = mid;
        }
        c = lo;
    }
    for (; c < idx->k && lt[c].off < i1; ++c) {
        const uint64_t lb = lt[c].off, le = c + 1 < idx->k ? lt[c + 1].off : idx->n;
        const uint64_t a = std::max<uint64_t>(lb, i0), e = std::min<uint64_t>(le, i1);
        if (a >= e) continue;
        const uint64_t split = lb + lt[c].h;  // positions [lb, split) in HBM, [split, le) on the host
        if (a < split) {
            const uint64_t e2 = std::min(e, split);
            RQC(dev_copy(lt[c].hbm_base + (a - lb), e2 - a, buf + (a - i0) * dim));
        }
        if (e > split) {
            const uint64_t a2 = std::max(a, split);
            float *hrow = idx->base_host + ((uint64_t)lt[c].host_base + (a2 - split)) * dim;
            if (to_index) memcpy(hrow, buf + (a2 - i0) * dim, (e - a2) * dim * 4);
            else memcpy(buf + (a2 - i0) * dim, hrow, (e - a2) * dim * 4);
        }
    }
    return RQ_OK;
}

// ------------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *rq_version(void) { return "rabitq_hip 0.4.0 (gfx950, abi 4)"; }
uint32_t rq_abi_version(void) { return RQ_ABI_VERSION; }
const char *rq_last_error(void) { return g_err.c_str(); }

rq_status rq_init(int device) {
    RQC(ensure_device());
    HIPC(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(RQ_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is gfx950-only");
    return RQ_OK;
}

rq_status rq_kmeans_device(const float *d_base, uint64_t n, uint32_t d, uint32_t k, uint32_t iters,
                           uint32_t points_per_centroid, uint64_t seed, float *d_centroids_out) {
    RQC(ensure_device());
    RQC(ensure_kernel_attributes());
    if (!d_base || !d_centroids_out || n == 0 || d == 0 || k == 0) return fail(RQ_ERR_INVALID, "bad k-means arguments");
    const uint32_t dim = (d + 63) / 64 * 64;
    if (dim > 4096) return fail(RQ_ERR_UNSUPPORTED, "dim > 4096 not supported");
    const uint64_t ns = std::max<uint64_t>(k, std::min<uint64_t>(n, (uint64_t)std::max(points_per_centroid, 1u) * k));
    if (ns * dim > (1ull << 31)) return fail(RQ_ERR_UNSUPPORTED, "k-means sample too large");
    rq_index tmp;  // only its centroid buffers are used (launch_assign)
    tmp.dim = dim, tmp.k = k, tmp.W = dim / 64;
    DevBuf<float> xs, sums;
    DevBuf<uint32_t> label, counts;
    DevBuf<float> dist;
    RQC(xs.alloc(ns * dim));
    RQC(tmp.centroids.alloc((size_t)k * dim));
    RQC(tmp.cent_t.alloc((size_t)k * dim));
    RQC(sums.alloc((size_t)k * dim));
    RQC(label.alloc(ns));
    RQC(dist.alloc(ns));
    RQC(counts.alloc(k));
    kmeans_sample_kernel<<<ceil_div(ns * dim, 256), 256>>>(d_base, n, d, dim, seed, ns, xs.p);
    HIPC(hipMemcpy(tmp.centroids.p, xs.p, (size_t)k * dim * 4, hipMemcpyDeviceToDevice));  // init: first k sample rows
    for (uint32_t it = 0; it < iters; ++it) {
        transpose_kernel<<<dim3(ceil_div(dim, 32), ceil_div(k, 32)), dim3(32, 8)>>>(tmp.centroids.p, tmp.cent_t.p, k, dim);
        for (uint64_t i0 = 0; i0 < ns; i0 += (1ull << 20)) {
            const uint64_t m = std::min<uint64_t>(1ull << 20, ns - i0);
            launch_assign(xs.p + i0 * dim, &tmp, m, label.p + i0, dist.p + i0, nullptr);
        }
        HIPC(hipMemset(sums.p, 0, (size_t)k * dim * 4));
        HIPC(hipMemset(counts.p, 0, (size_t)k * 4));
        kmeans_accumulate_kernel<<<ceil_div(ns * dim, 256), 256>>>(xs.p, label.p, ns, dim, sums.p, counts.p);
        kmeans_update_kernel<<<ceil_div((uint64_t)k * dim, 256), 256>>>(sums.p, counts.p, xs.p, ns, k, dim,
                                                                        seed + it + 1, tmp.centroids.p);
    }
    unpad_rows_kernel<<<ceil_div((uint64_t)k * d, 256), 256>>>(tmp.centroids.p, d_centroids_out, k, dim, d);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    return RQ_OK;
}

rq_status rq_build_device(const float *d_base, uint64_t n, uint32_t d, const float *d_centroids, uint32_t k,
                          const float *orthogonal_host, uint64_t seed, rq_index **out) {
    return build_device(d_base, n, d, d_centroids, k, orthogonal_host, seed, out);
}

rq_status rq_builder_create(uint64_t n, uint32_t d, const float *d_centroids, uint32_t k, const float *orthogonal_host,
                            uint64_t seed, uint64_t max_device_base_bytes, rq_builder **out) {
    return builder_create(n, d, d_centroids, k, orthogonal_host, seed, max_device_base_bytes, out);
}
rq_status rq_builder_assign_chunk(rq_builder *b, const float *d_rows, uint64_t i0, uint64_t m) { return builder_assign(b, d_rows, i0, m); }
rq_status rq_builder_order(rq_builder *b) { return builder_order(b); }
rq_status rq_builder_place_chunk(rq_builder *b, const float *d_rows, uint64_t i0, uint64_t m) { return builder_place(b, d_rows, i0, m); }
rq_status rq_builder_finish(rq_builder *b, rq_index **out) { return builder_finish(b, out); }
void rq_builder_free(rq_builder *b) { delete b; }
rq_status rq_builder_stats(const rq_builder *b, rq_build_stats_t *out) {
    if (!b || !out) return fail(RQ_ERR_INVALID, "null argument");
    return copy_out_sized(out, b->stats);
}

rq_status rq_build(const float *base, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                   const float *orthogonal, uint64_t seed, rq_index **out) {
    RQC(ensure_device());
    if ((n && !base) || !centroids) return fail(RQ_ERR_INVALID, "null argument");
    DevBuf<float> db, dc;
    RQC(db.alloc(n * d));
    RQC(dc.alloc((size_t)k * d));
    if (n) HIPC(hipMemcpy(db.p, base, n * d * 4, hipMemcpyHostToDevice));
    if (k) HIPC(hipMemcpy(dc.p, centroids, (size_t)k * d * 4, hipMemcpyHostToDevice));
    return build_device(db.p, n, d, dc.p, k, orthogonal, seed, out);
}

rq_status rq_build_from_path(const char *base_fvecs, const char *centroid_fvecs, const float *orthogonal,
                             uint64_t seed, rq_index **out) {
    if (!base_fvecs || !centroid_fvecs) return fail(RQ_ERR_INVALID, "null path");
    VecsFile b, c;
    RQC(read_vecs_file(base_fvecs, 4, b));      // rabitq.rs:160
    RQC(read_vecs_file(centroid_fvecs, 4, c));  // :163
    if (b.lens.empty() || c.lens.empty()) return fail(RQ_ERR_IO, "empty fvecs file");
    uint32_t d = b.lens[0];
    if (c.lens[0] != d) return fail(RQ_ERR_DIM_MISMATCH, "base and centroid dimensions differ (rabitq.rs:165)");
    for (uint32_t l : b.lens)
        if (l != d) return fail(RQ_ERR_IO, "ragged base.fvecs");
    for (uint32_t l : c.lens)
        if (l != d) return fail(RQ_ERR_IO, "ragged centroids.fvecs");
    return rq_build(reinterpret_cast<const float *>(b.data.data()), b.lens.size(), d,
                    reinterpret_cast<const float *>(c.data.data()), (uint32_t)c.lens.size(), orthogonal, seed, out);
}

rq_status rq_from_arrays(uint32_t dim, uint64_t n, uint32_t k, const float *base, const float *orthogonal,
                         const float *centroids, const uint32_t *offsets, const uint32_t *map_ids,
                         const uint64_t *codes, const rq_factor_t *factors, rq_index **out) {
    return from_arrays(dim, n, k, base, orthogonal, centroids, offsets, map_ids, codes, factors, out);
}

// rabitq.rs:84-125
rq_status rq_load_dir(const char *dir, rq_index **out) {
    if (!dir || !out) return fail(RQ_ERR_INVALID, "null argument");
    const std::string d(dir);
    VecsFile ortho, cent, oi, fac, bin, base;
    RQC(read_vecs_file(d + "/orthogonal.fvecs", 4, ortho));
    RQC(read_vecs_file(d + "/centroids.fvecs", 4, cent));
    RQC(read_vecs_file(d + "/offsets_ids.ivecs", 4, oi));
    RQC(read_vecs_file(d + "/factors.fvecs", 4, fac));
    RQC(read_vecs_file(d + "/x_binary_vec.u64vecs", 8, bin));
    RQC(read_vecs_file(d + "/base.fvecs", 4, base));
    const uint32_t dim = (uint32_t)ortho.lens.size();  // :108 dim = orthogonal.nrows()
    if (dim == 0 || dim % 64 != 0) return fail(RQ_ERR_DIM_MISMATCH, "orthogonal.fvecs: dim % 64 != 0 (rabitq.rs:109)");
    if (cent.lens.size() != dim || oi.lens.size() != 2) return fail(RQ_ERR_IO, "malformed index directory");
    const uint32_t k = cent.lens[0];
    // every record length is checked before anything is indexed by it (the reference's matrix_from_fvecs panics on
    // ragged input, src/utils.rs:44-49)
    for (uint32_t l : ortho.lens)
        if (l != dim) return fail(RQ_ERR_IO, "orthogonal.fvecs is not dim x dim");
    for (uint32_t l : cent.lens)
        if (l != k) return fail(RQ_ERR_IO, "centroids.fvecs is not dim records of k values");
    for (uint32_t l : base.lens)
        if (l != dim) return fail(RQ_ERR_IO, "base.fvecs record length != dim");
    // centroids.fvecs holds the dim x k matrix row-wise: un-transpose to k x dim
    std::vector<float> c((size_t)k * dim);
    const float *ct = reinterpret_cast<const float *>(cent.data.data());
    for (uint32_t r = 0; r < dim; ++r)
        for (uint32_t j = 0; j < k; ++j) c[(size_t)j * dim + r] = ct[(size_t)r * k + j];
    const uint32_t *oip = reinterpret_cast<const uint32_t *>(oi.data.data());
    const uint32_t first = oi.lens.front(), last = oi.lens.back();
    size_t total = 0;
    for (uint32_t l : oi.lens) total += l;
    if (first != k + 1) return fail(RQ_ERR_IO, "offsets record length != k + 1");
    const uint64_t n = last;
    if (fac.data.size() != n * 16 || bin.data.size() != n * (dim / 64) * 8 || base.data.size() != n * dim * 4 ||
        base.lens.size() != n)
        return fail(RQ_ERR_IO, "index arrays disagree on n");
    if (oip[k] != n) return fail(RQ_ERR_IO, "offsets[k] != number of vectors");
    for (uint32_t j = 0; j < k; ++j)
        if (oip[j] > oip[j + 1]) return fail(RQ_ERR_IO, "offsets are not non-decreasing");
    return from_arrays(dim, n, k, reinterpret_cast<const float *>(base.data.data()),
                       reinterpret_cast<const float *>(ortho.data.data()), c.data(), oip, oip + (total - last),
                       reinterpret_cast<const uint64_t *>(bin.data.data()),
                       reinterpret_cast<const rq_factor_t *>(fac.data.data()), out);
}

rq_status rq_get_array(const rq_index *idx, int which, void *dst, uint64_t dst_bytes);

// rabitq.rs:128-156
rq_status rq_dump_dir(const rq_index *idx, const char *dir) {
    if (!idx || !dir) return fail(RQ_ERR_INVALID, "null argument");
    mkdir(dir, 0777);
    const std::string d(dir);
    const uint32_t dim = idx->dim, k = idx->k;
    const uint64_t n = idx->n;
    if (4 * n > 0xFFFFFFFFull || n * (dim / 64) > 0xFFFFFFFFull)
        return fail(RQ_ERR_UNSUPPORTED, "single-record files need 4n and n*dim/64 to fit the u32 header (rabitq.rs:141-155)");
    auto open = [&](const char *name, FILE **f) -> rq_status {
        *f = fopen((d + "/" + name).c_str(), "wb");
        return *f ? RQ_OK : fail(RQ_ERR_IO, "cannot create " + d + "/" + name);
    };
    FILE *f;
    {  // base.fvecs: n records of dim (cluster order), streamed in chunks
        RQC(open("base.fvecs", &f));
        const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / (dim * 4));
        std::vector<float> buf(std::min<uint64_t>(chunk, std::max<uint64_t>(n, 1)) * dim);
        for (uint64_t i0 = 0; i0 < n; i0 += chunk) {
            uint64_t m = std::min(chunk, n - i0);
            if (copy_base_rows(idx, i0, m, buf.data(), false) != RQ_OK) {
                fclose(f);
                return RQ_ERR_HIP;
            }
            for (uint64_t i = 0; i < m; ++i) {
                rq_status s = write_record(f, buf.data() + i * dim, dim, 4, "base.fvecs");
                if (s != RQ_OK) {
                    fclose(f);
                    return s;
                }
            }
        }
        fclose(f);
    }
    std::vector<float> P((size_t)dim * dim), C((size_t)k * dim), row(std::max<uint32_t>(k, 1));
    std::vector<uint32_t> off((size_t)k + 1), ids(n);
    std::vector<float> fac(n * 4);
    std::vector<uint64_t> codes(n * (dim / 64));
    RQC(rq_get_array(idx, RQ_ARR_ORTHOGONAL, P.data(), P.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_CENTROIDS, C.data(), C.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_OFFSETS, off.data(), off.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_MAP_IDS, ids.data(), ids.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_FACTORS, fac.data(), fac.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_CODES, codes.data(), codes.size() * 8));
    rq_status s = RQ_OK;
    RQC(open("orthogonal.fvecs", &f));
    for (uint32_t r = 0; r < dim && s == RQ_OK; ++r) s = write_record(f, P.data() + (size_t)r * dim, dim, 4, "orthogonal.fvecs");
    fclose(f);
    RQC(s);
    RQC(open("centroids.fvecs", &f));  // dim records of k values (rotated, transposed)
    for (uint32_t r = 0; r < dim && s == RQ_OK; ++r) {
        for (uint32_t j = 0; j < k; ++j) row[j] = C[(size_t)j * dim + r];
        s = write_record(f, row.data(), k, 4, "centroids.fvecs");
    }
    fclose(f);
    RQC(s);
    RQC(open("offsets_ids.ivecs", &f));
    s = write_record(f, off.data(), k + 1, 4, "offsets_ids.ivecs");
    if (s == RQ_OK) s = write_record(f, ids.data(), (uint32_t)n, 4, "offsets_ids.ivecs");
    fclose(f);
    RQC(s);
    RQC(open("factors.fvecs", &f));
    s = write_record(f, fac.data(), (uint32_t)(4 * n), 4, "factors.fvecs");
    fclose(f);
    RQC(s);
    RQC(open("x_binary_vec.u64vecs", &f));
    s = write_record(f, codes.data(), (uint32_t)(n * (dim / 64)), 8, "x_binary_vec.u64vecs");
    fclose(f);
    return s;
}

// ---- JSON persistence: load_from_json / dump_to_json, src/rabitq.rs:72-81 ------------------------------------
// serde_json of `RaBitQ` (src/rabitq.rs:56-68): {"dim", "base": Mat, "orthogonal": Mat, "centroids": Mat, "rand_bias",
// "offsets", "map_ids", "x_binary_vec", "factors": [{factor_ip, factor_ppc, error_bound, center_distance_square}]}.
// A faer 0.19 `Mat` serialises as {"nrows", "ncols", "data": row-major sequence}; base is dim x n (one vector per
// column, :188), centroids dim x k (:189).  faer's source is not in the reference tree, so the Mat layout is restated
// from its published serde impl (parity unpinned, like every faer call site: DESIGN.md section 2).  Numbers are
// written in shortest round-trip form; f32 non-finite values become null as serde_json writes them (and, as in
// serde_json, a null does not load).  rand_bias is not used by the AVX2 path (src/simd.rs:177): 0.5 per dimension.

rq_status rq_dump_json(const rq_index *idx, const char *path) {
    if (!idx || !path) return fail(RQ_ERR_INVALID, "null argument");
    const uint64_t n = idx->n, dim = idx->dim, k = idx->k;
    std::vector<float> base(n * dim), P(dim * dim), C(k * dim), fac(n * 4);
    std::vector<uint32_t> off(k + 1), ids(n);
    std::vector<uint64_t> codes(n * idx->W);
    RQC(rq_get_array(idx, RQ_ARR_BASE, base.data(), base.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_ORTHOGONAL, P.data(), P.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_CENTROIDS, C.data(), C.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_OFFSETS, off.data(), off.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_MAP_IDS, ids.data(), ids.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_FACTORS, fac.data(), fac.size() * 4));
    RQC(rq_get_array(idx, RQ_ARR_CODES, codes.data(), codes.size() * 8));
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RQ_ERR_IO, std::string("cannot create ") + path);
    JsonOut o{f};
    o.raw("{\"dim\":"), o.u64(dim);
    o.raw(",\"base\":"), o.mat(base.data(), dim, n, true);          // dim x n: column j = vector j
    o.raw(",\"orthogonal\":"), o.mat(P.data(), dim, dim, false);
    o.raw(",\"centroids\":"), o.mat(C.data(), dim, k, true);       // dim x k: column j = rotated centroid j
    o.raw(",\"rand_bias\":[");
    for (uint64_t i = 0; i < dim; ++i) o.raw(i ? ",0.5" : "0.5");
    o.raw("],\"offsets\":[");
    for (uint64_t i = 0; i <= k; ++i) o.raw(i ? "," : ""), o.u64(off[i]);
    o.raw("],\"map_ids\":[");
    for (uint64_t i = 0; i < n; ++i) o.raw(i ? "," : ""), o.u64(ids[i]);
    o.raw("],\"x_binary_vec\":[");
    for (uint64_t i = 0; i < codes.size(); ++i) o.raw(i ? "," : ""), o.u64(codes[i]);
    o.raw("],\"factors\":[");
    for (uint64_t i = 0; i < n; ++i) {
        o.raw(i ? ",{\"factor_ip\":" : "{\"factor_ip\":"), o.f32(fac[4 * i]);
        o.raw(",\"factor_ppc\":"), o.f32(fac[4 * i + 1]);
        o.raw(",\"error_bound\":"), o.f32(fac[4 * i + 2]);
        o.raw(",\"center_distance_square\":"), o.f32(fac[4 * i + 3]), o.raw("}");
    }
    o.raw("]}");
    const bool closed = fclose(f) == 0;
    if (!o.ok || !closed) return fail(RQ_ERR_IO, std::string("write error on ") + path);
    return RQ_OK;
}

rq_status rq_load_json(const char *path, rq_index **out) {
    if (!path || !out) return fail(RQ_ERR_INVALID, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(RQ_ERR_IO, std::string("cannot open ") + path);  // "open json error"
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::string text((size_t)std::max(sz, 0l), '\0');
    const bool read_ok = sz <= 0 || fread(&text[0], 1, (size_t)sz, f) == (size_t)sz;
    fclose(f);
    if (!read_ok) return fail(RQ_ERR_IO, std::string("short read on ") + path);
    JsonIn in{text.data(), text.data() + text.size(), {}};
    unsigned long long dim = 0, br = 0, bc = 0, pr = 0, pc = 0, cr = 0, cc = 0;
    std::vector<float> base, P, cent, bias;
    std::vector<unsigned long long> off, ids, codes;
    std::vector<rq_factor_t> fac;
    bool good = in.need('{');
    if (good && !in.lit('}')) {
        do {
            std::string k;
            if (!(good = in.key(k))) break;
            if (k == "dim") good = in.num_u64(dim);
            else if (k == "base") good = in.mat(base, br, bc);
            else if (k == "orthogonal") good = in.mat(P, pr, pc);
            else if (k == "centroids") good = in.mat(cent, cr, cc);
            else if (k == "rand_bias") good = in.array(bias, [&](float &v) { return in.num_f32(v); });
            else if (k == "offsets") good = in.array(off, [&](unsigned long long &v) { return in.num_u64(v); });
            else if (k == "map_ids") good = in.array(ids, [&](unsigned long long &v) { return in.num_u64(v); });
            else if (k == "x_binary_vec") good = in.array(codes, [&](unsigned long long &v) { return in.num_u64(v); });
            else if (k == "factors")
                good = in.array(fac, [&](rq_factor_t &fv) {
                    fv = rq_factor_t{0, 0, 0, 0};
                    if (!in.need('{')) return false;
                    do {
                        std::string fk;
                        if (!in.key(fk)) return false;
                        float *dst = fk == "factor_ip" ? &fv.factor_ip : fk == "factor_ppc" ? &fv.factor_ppc
                                   : fk == "error_bound" ? &fv.error_bound
                                   : fk == "center_distance_square" ? &fv.center_distance_square : nullptr;
                        if (dst ? !in.num_f32(*dst) : !in.skip()) return false;
                    } while (in.lit(','));
                    return in.need('}');
                });
            else good = in.skip();
        } while (good && in.lit(','));
        good = good && in.need('}');
    }
    if (!good) return fail(RQ_ERR_IO, std::string("deserialize error in ") + path + (in.err.empty() ? "" : ": " + in.err));
    const uint64_t n = ids.size(), k = off.empty() ? 0 : off.size() - 1;
    if (dim == 0 || dim % 64 || pr != dim || pc != dim || P.size() != dim * dim || br != dim || bc != n || base.size() != dim * n ||
        cr != dim || cc != k || cent.size() != dim * k || off.empty() || fac.size() != n || codes.size() != n * (dim / 64) ||
        off.back() != n)
        return fail(RQ_ERR_IO, std::string("inconsistent index in ") + path);
    for (uint64_t j = 0; j + 1 < off.size(); ++j)
        if (off[j] > off[j + 1]) return fail(RQ_ERR_IO, "offsets are not non-decreasing");
    // Mat (dim x cols, row-major sequence) -> one vector per row
    std::vector<float> base_rows(n * dim), cent_rows(k * dim);
    for (uint64_t i = 0; i < dim; ++i) {
        for (uint64_t j = 0; j < n; ++j) base_rows[j * dim + i] = base[i * n + j];
        for (uint64_t j = 0; j < k; ++j) cent_rows[j * dim + i] = cent[i * k + j];
    }
    std::vector<uint32_t> off32(off.begin(), off.end()), ids32(ids.begin(), ids.end());
    std::vector<uint64_t> codes64(codes.begin(), codes.end());
    return from_arrays((uint32_t)dim, n, (uint32_t)k, base_rows.data(), P.data(), cent_rows.data(), off32.data(), ids32.data(),
                       codes64.data(), fac.data(), out);
}

void rq_free(rq_index *idx) { delete idx; }

rq_status rq_info(const rq_index *idx, rq_info_t *out) {
    if (!idx || !out) return fail(RQ_ERR_INVALID, "null argument");
    rq_info_t full{};
    full.dim = idx->dim, full.k = idx->k, full.n = idx->n, full.max_list_len = idx->max_list_len, full.n_hbm = idx->n_dev;
    return copy_out_sized(out, full);
}

rq_status rq_get_device_ptr(const rq_index *idx, int which, const void **out_ptr, uint64_t *out_bytes) {
    if (!idx || !out_ptr || !out_bytes) return fail(RQ_ERR_INVALID, "null argument");
    switch (which) {
        case RQ_ARR_BASE:
            if (idx->n_dev < idx->n)  // tiered: the HBM tier holds packed list heads, not rows at their positions
                return fail(RQ_ERR_UNSUPPORTED, "the raw vectors of this index are tiered (HBM + pinned host memory): no single device array; use rq_get_array");
            *out_ptr = idx->base.p, *out_bytes = idx->n_dev * idx->dim * 4;
            break;
        case RQ_ARR_ORTHOGONAL: *out_ptr = idx->P.p, *out_bytes = (uint64_t)idx->dim * idx->dim * 4; break;
        case RQ_ARR_CENTROIDS: *out_ptr = idx->centroids.p, *out_bytes = (uint64_t)idx->k * idx->dim * 4; break;
        case RQ_ARR_OFFSETS: *out_ptr = idx->offsets.p, *out_bytes = ((uint64_t)idx->k + 1) * 4; break;
        case RQ_ARR_MAP_IDS: *out_ptr = idx->map_ids.p, *out_bytes = idx->n * 4; break;
        case RQ_ARR_CODES: *out_ptr = idx->codes.p, *out_bytes = idx->n * idx->W * 8; break;
        case RQ_ARR_FACTORS: *out_ptr = idx->factors.p, *out_bytes = idx->n * 16; break;
        default: return fail(RQ_ERR_INVALID, "unknown array id");
    }
    return RQ_OK;
}

rq_status rq_get_array(const rq_index *idx, int which, void *dst, uint64_t dst_bytes) {
    const void *p;
    uint64_t bytes;
    if (idx && which == RQ_ARR_BASE && idx->n_dev < idx->n) {  // both tiers
        if (!dst || dst_bytes < idx->n * idx->dim * 4) return fail(RQ_ERR_INVALID, "destination too small");
        return copy_base_rows(idx, 0, idx->n, static_cast<float *>(dst), false);
    }
    RQC(rq_get_device_ptr(idx, which, &p, &bytes));
    if (!dst || dst_bytes < bytes) return fail(RQ_ERR_INVALID, "destination too small");
    if (bytes) HIPC(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    return RQ_OK;
}

rq_status rq_coarse_topk_device(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                uint32_t list_lo, uint32_t list_hi, uint32_t probe, uint32_t *d_out_cluster,
                                float *d_out_dist) {
    RQC(ensure_device());
    if (!idx || !d_queries || !d_out_cluster || !d_out_dist) return fail(RQ_ERR_INVALID, "null argument");
    if (idx->dim != (len + 63) / 64 * 64) return fail(RQ_ERR_DIM_MISMATCH, "query length does not pad to dim");
    if (probe == 0 || list_lo >= list_hi || list_hi > idx->k) return fail(RQ_ERR_INVALID, "bad list range / probe");
    if (probe > RQ_MAX_PROBE) return fail(RQ_ERR_UNSUPPORTED, "probe > 16384");
    if (nq == 0) return RQ_OK;
    const uint32_t dim = idx->dim, kc = list_hi - list_lo, np = std::min(probe, kc);
    // runs on a pooled workspace (its buffers and stream): this entry sits in the per-batch loop of sharded
    // deployments, so nothing may be allocated or freed per call
    rq_index *mi = const_cast<rq_index *>(idx);
    Workspace *ws = ws_acquire(mi);
    struct Rel {
        rq_index *i;
        Workspace *w;
        ~Rel() { ws_release(i, w); }
    } rel{mi, ws};
    if (!ws->stream) HIPC(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));
    hipStream_t st = ws->stream;
    // the distance matrix is chunk x kc floats: queries go through in chunks of at most 2^31 cells (8 GiB), so that a
    // multi-GPU batch (65536 x N queries) against tens of thousands of lists does not ask for one 70 GB buffer
    const uint32_t chunk = (uint32_t)std::min<uint64_t>(nq, std::max<uint64_t>(1024, (1ull << 31) / std::max(kc, idx->k)));
    RQC(ws->y.ensure((uint64_t)chunk * dim));
    RQC(ws->dist.ensure((uint64_t)chunk * std::max(kc, idx->k)));
    if (len != dim) RQC(ws->qpad.ensure((uint64_t)chunk * dim));
    for (uint32_t q0 = 0; q0 < nq; q0 += chunk) {
        const uint32_t m = std::min(chunk, nq - q0);
        const float *qp = d_queries + (uint64_t)q0 * len;
        if (len != dim) {
            pad_rows_kernel<<<ceil_div((uint64_t)m * dim, 256), 256, 0, st>>>(qp, ws->qpad.p, m, len, dim);
            qp = ws->qpad.p;
        }
        launch_rotate(qp, idx->P.p, ws->y.p, m, dim, m >= 32, st);
        if (kc == idx->k && coarse_prefilter_applies(idx, m, np)) {
            RQC(ws->coarse_redo.ensure(m));
            RQC(ws->qf6.ensure((size_t)m * dim / 2 + 16));
            launch_coarse_prefiltered(idx, ws->y.p, ws->dist.p, m, np, d_out_cluster + (uint64_t)q0 * probe, d_out_dist + (uint64_t)q0 * probe, probe, nullptr,
                                      ws->coarse_redo.p, st, reinterpret_cast<uint16_t *>(ws->qf6.p));
            continue;
        }
        launch_coarse(idx->cent_t.p + list_lo, ws->y.p, ws->dist.p, kc, dim, m, idx->k, st);
        launch_select(ws->dist.p, kc, np, d_out_cluster + (uint64_t)q0 * probe, d_out_dist + (uint64_t)q0 * probe, list_lo, probe, m, st);
    }
    HIPC(hipStreamSynchronize(st));
    HIPC(hipGetLastError());
    return RQ_OK;
}

rq_status rq_merge_smallest_u64_device(const uint64_t *d_in, uint32_t world, uint32_t nq, uint32_t width, uint32_t m_out,
                                       uint64_t *d_out) {
    RQC(ensure_device());
    if (!d_in || !d_out) return fail(RQ_ERR_INVALID, "null argument");
    if (nq == 0 || m_out == 0) return RQ_OK;
    const uint64_t m = (uint64_t)world * width;
    if (m == 0 || m > 16384) return fail(RQ_ERR_UNSUPPORTED, "world * width must be in [1, 16384]");
    RQC(ensure_kernel_attributes());
    // on the legacy default stream, asynchronously: ordered with the caller's default-stream work (torch's
    // current stream is that stream unless the caller changed it), like a library call of its own framework
    merge_smallest_u64_kernel<<<nq, 256, (size_t)pow2_ceil((uint32_t)m) * 8, nullptr>>>(
        reinterpret_cast<const unsigned long long *>(d_in), world, nq, width, m_out, reinterpret_cast<unsigned long long *>(d_out),
        (uint64_t)nq * width);
    HIPC(hipGetLastError());
    return RQ_OK;
}

rq_status rq_query_batch_device_probed(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                       const uint32_t *d_probe_cluster, const float *d_probe_dist, uint32_t probe,
                                       uint32_t topk, int heuristic_rank, float *d_out_dist, uint32_t *d_out_id,
                                       uint32_t *d_out_n) {
    if (!d_probe_cluster || !d_probe_dist) return fail(RQ_ERR_INVALID, "null probe lists");
    if (idx && probe > idx->k) return fail(RQ_ERR_INVALID, "probe lists must have min(probe, k) columns: pass probe <= k");
    return query_device(const_cast<rq_index *>(idx), d_queries, nq, len, probe, topk, heuristic_rank != 0, d_out_dist,
                        d_out_id, d_out_n, d_probe_cluster, d_probe_dist);
}

rq_status rq_query_batch_device_seeded(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                       const uint32_t *d_probe_cluster, const float *d_probe_dist, uint32_t probe,
                                       uint32_t topk, int heuristic_rank, const float *d_thr_init, float *d_out_dist,
                                       uint32_t *d_out_id, uint32_t *d_out_n) {
    if (!d_probe_cluster || !d_probe_dist || !d_thr_init) return fail(RQ_ERR_INVALID, "null probe lists / thresholds");
    if (idx && probe > idx->k) return fail(RQ_ERR_INVALID, "probe lists must have min(probe, k) columns: pass probe <= k");
    return query_device(const_cast<rq_index *>(idx), d_queries, nq, len, probe, topk, heuristic_rank != 0, d_out_dist,
                        d_out_id, d_out_n, d_probe_cluster, d_probe_dist, nullptr, d_thr_init);
}

rq_status rq_query_batch_device(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                uint32_t probe, uint32_t topk, int heuristic_rank, float *d_out_dist,
                                uint32_t *d_out_id, uint32_t *d_out_n) {
    return query_device(const_cast<rq_index *>(idx), d_queries, nq, len, probe, topk, heuristic_rank != 0, d_out_dist,
                        d_out_id, d_out_n);
}

rq_status rq_query_batch_device_begin(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                      uint32_t probe, uint32_t topk, int heuristic_rank, float *d_out_dist,
                                      uint32_t *d_out_id, uint32_t *d_out_n, rq_ticket **out_ticket) {
    return query_device_begin(const_cast<rq_index *>(idx), d_queries, nq, len, probe, topk, heuristic_rank != 0,
                              d_out_dist, d_out_id, d_out_n, out_ticket);
}
rq_status rq_query_batch_device_end(rq_ticket *ticket) { return query_device_end(ticket); }

// Device staging of the host-pointer entry points: grown on demand, kept per host thread so a
// per-vector `query()` loop does not pay hipMalloc/hipFree on every call (intentionally never freed).
struct HostStaging {
    void *p[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[4] = {0, 0, 0, 0};
    void *pin = nullptr;  // pinned, device-mapped host buffer for small calls (zero-copy in both directions)
    size_t pin_cap = 0;
    rq_status get(int i, size_t bytes, void **out) {
        if (bytes > cap[i]) {
            if (p[i]) (void)hipFree(p[i]);
            p[i] = nullptr, cap[i] = 0;
            size_t want = std::max<size_t>(bytes, 4096);
            hipError_t e = hipMalloc(&p[i], want);
            if (e != hipSuccess) return fail(RQ_ERR_OOM, std::string("staging hipMalloc failed: ") + hipGetErrorString(e));
            cap[i] = want;
        }
        *out = p[i];
        return RQ_OK;
    }
    rq_status get_pinned(size_t bytes, void **out) {
        if (bytes > pin_cap) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr, pin_cap = 0;
            size_t want = std::max<size_t>(bytes, 65536);
            hipError_t e = hipHostMalloc(&pin, want, hipHostMallocMapped | hipHostMallocCoherent);
            if (e != hipSuccess) return fail(RQ_ERR_OOM, std::string("pinned staging allocation failed: ") + hipGetErrorString(e));
            pin_cap = want;
        }
        *out = pin;
        return RQ_OK;
    }
};
static thread_local HostStaging g_staging;
#define RQ_ZERO_COPY_BYTES (1u << 20)

rq_status rq_query_batch(const rq_index *idx, const float *queries, uint32_t nq, uint32_t len, uint32_t probe,
                         uint32_t topk, int heuristic_rank, float *out_dist, uint32_t *out_id, uint32_t *out_n) {
    RQC(ensure_device());
    if (!idx || !queries || !out_dist || !out_id || !out_n) return fail(RQ_ERR_INVALID, "null argument");
    if (nq == 0) return RQ_OK;
    if (topk == 0) return fail(RQ_ERR_UNSUPPORTED, "topk must be in [1, 2048]");
    const size_t qb = ((size_t)nq * len * 4 + 255) & ~(size_t)255, ob = ((size_t)nq * topk * 4 + 255) & ~(size_t)255;
    if (qb + 2 * ob + (size_t)nq * 4 <= RQ_ZERO_COPY_BYTES) {
        // small call (the reference's one-query-per-call loop): queries and results live in pinned host memory
        // the kernels address directly, which replaces five blocking copies by two host memcpys
        void *pin;
        RQC(g_staging.get_pinned(qb + 2 * ob + (size_t)nq * 4, &pin));
        void *dev = nullptr;
        HIPC(hipHostGetDevicePointer(&dev, pin, 0));
        char *h = static_cast<char *>(pin), *d = static_cast<char *>(dev);
        memcpy(h, queries, (size_t)nq * len * 4);
        rq_status s = query_device(const_cast<rq_index *>(idx), (const float *)d, nq, len, probe, topk, heuristic_rank != 0,
                                   (float *)(d + qb), (uint32_t *)(d + qb + ob), (uint32_t *)(d + qb + 2 * ob));
        if (s != RQ_OK && s != RQ_ERR_EMPTY) return s;
        memcpy(out_dist, h + qb, (size_t)nq * topk * 4);
        memcpy(out_id, h + qb + ob, (size_t)nq * topk * 4);
        memcpy(out_n, h + qb + 2 * ob, (size_t)nq * 4);
        return s;
    }
    void *dq, *dd, *di, *dn;
    RQC(g_staging.get(0, (uint64_t)nq * len * 4, &dq));
    RQC(g_staging.get(1, (uint64_t)nq * topk * 4, &dd));
    RQC(g_staging.get(2, (uint64_t)nq * topk * 4, &di));
    RQC(g_staging.get(3, (uint64_t)nq * 4, &dn));
    HIPC(hipMemcpy(dq, queries, (uint64_t)nq * len * 4, hipMemcpyHostToDevice));
    rq_status s = query_device(const_cast<rq_index *>(idx), (const float *)dq, nq, len, probe, topk, heuristic_rank != 0,
                               (float *)dd, (uint32_t *)di, (uint32_t *)dn);
    if (s != RQ_OK && s != RQ_ERR_EMPTY) return s;
    HIPC(hipMemcpy(out_dist, dd, (uint64_t)nq * topk * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(out_id, di, (uint64_t)nq * topk * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(out_n, dn, nq * 4, hipMemcpyDeviceToHost));
    return s;
}

rq_status rq_query(const rq_index *idx, const float *query, uint32_t len, uint32_t probe, uint32_t topk,
                   int heuristic_rank, float *out_dist, uint32_t *out_id, uint32_t *out_n) {
    return rq_query_batch(idx, query, 1, len, probe, topk, heuristic_rank, out_dist, out_id, out_n);
}

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

rq_status rq_metrics(rq_metrics_t *out) {
    if (!out) return fail(RQ_ERR_INVALID, "null argument");
    out->rough = g_rough.load(), out->precise = g_precise.load(), out->query = g_query.load(), out->miss = g_miss.load();
    return RQ_OK;
}
rq_status rq_metrics_reset(void) {
    g_rough = 0, g_precise = 0, g_query = 0, g_miss = 0;
    return RQ_OK;
}

rq_status rq_set_option(const char *name, int value) {
    if (!name) return fail(RQ_ERR_INVALID, "null option name");
    if (std::string(name) == "scan_impl") {
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "scan_impl must be 0 (auto), 1 (valu) or 2 (mfma)");
        g_scan_impl = value;
        return RQ_OK;
    }
    if (std::string(name) == "cluster_major_div") {  // developer knob (results identical for every value)
        if (value < 1 || value > 4096) return fail(RQ_ERR_INVALID, "cluster_major_div must be in [1, 4096]");
        g_cluster_major_div = value;
        return RQ_OK;
    }
    if (std::string(name) == "large_batch_from") {  // developer knob (results identical for every value): queries from which a batch takes the large-batch form of the stages
        if (value < 2 || value > (1 << 20)) return fail(RQ_ERR_INVALID, "large_batch_from must be in [2, 2^20]");
        g_large_from = value;
        return RQ_OK;
    }
    if (std::string(name) == "stage_settle_pct") {  // developer knob (results identical for every value): end of the early stages of a large batch, percent of the average list length
        if (value < 1 || value > 400) return fail(RQ_ERR_INVALID, "stage_settle_pct must be in [1, 400]");
        g_stage_settle_pct = value;
        return RQ_OK;
    }
    if (std::string(name) == "stage_growth") {  // geometric growth of the early stages (0 = default: 8, or 16 for small batches)
        if (value != 0 && (value < 2 || value > 64)) return fail(RQ_ERR_INVALID, "stage_growth must be 0 or in [2, 64]");
        g_stage_growth = value;
        return RQ_OK;
    }
    if (std::string(name) == "base_device_mb") {  // HBM budget of the raw vectors for indexes built / loaded from now on:
                                                  // -1 = automatic (default), else MiB; the rest goes to pinned host memory
        if (value < -1) return fail(RQ_ERR_INVALID, "base_device_mb must be >= -1");
        g_base_device_mb = value;
        return RQ_OK;
    }
    if (std::string(name) == "scan_tile_table") {  // full-list stages launch one block per existing (list, tile): 0 never, 1 auto, 2 always
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "scan_tile_table must be 0 (never), 1 (auto) or 2 (always)");
        g_scan_tile_table = value;
        return RQ_OK;
    }
    if (std::string(name) == "max_scan_blocks") {  // test hook: blocks per scan launch (0 = the hardware bound), forces chunked stages
        if (value < 0) return fail(RQ_ERR_INVALID, "max_scan_blocks must be >= 0");
        g_max_scan_blocks = value == 0 ? RQ_MAX_BLOCKS_256 : std::min<uint32_t>((uint32_t)value, RQ_MAX_BLOCKS_256);
        return RQ_OK;
    }
    if (std::string(name) == "pair_split") {  // test hook: 1 (default) = sharded passes list their non-empty pairs before the query quantisation, 0 = never
        if (value < 0 || value > 1) return fail(RQ_ERR_INVALID, "pair_split must be 0 or 1");
        g_pair_split = value;
        return RQ_OK;
    }
    if (std::string(name) == "coarse_tiled_from") {  // developer knob: list count from which the pre-filtered ranking selects through tile minima
        if (value < 0) return fail(RQ_ERR_INVALID, "coarse_tiled_from must be >= 0");
        g_coarse_tiled_from = value;
        return RQ_OK;
    }
    if (std::string(name) == "coarse_impl") {  // test hook: coarse-distance kernel (0 auto, 1 LDS broadcast, 2 scalar registers)
        if (value < 0 || value > 4) return fail(RQ_ERR_INVALID, "coarse_impl must be 0, 1, 2, 3 or 4");
        g_coarse_impl = value;
        return RQ_OK;
    }
    if (std::string(name) == "survivor_segments") {  // 0 never, 1 automatic (default), 2 every batch of >= 256 queries (tests); developer build: 3 = 2 with every arena stage failing (the fall-back)
#ifdef RQ_DEV_ABLATIONS
        if (value < 0 || value > 3) return fail(RQ_ERR_INVALID, "survivor_segments must be 0, 1, 2 or 3");
#else
        if (value == 3) return fail(RQ_ERR_INVALID, "survivor_segments = 3 (allocation-failure injection) exists in the developer build only (make dev)");
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "survivor_segments must be 0, 1 or 2");
#endif
        g_seg_opt = value;
        return RQ_OK;
    }
    if (std::string(name) == "assign_impl") {  // nearest-list assignment of builds started from now on: 0 = matrix-core pre-filter + exact refinement, 1 = exact-order VALU kernels only
        if (value < 0 || value > 1) return fail(RQ_ERR_INVALID, "assign_impl must be 0 or 1");
        g_assign_impl = value;
        return RQ_OK;
    }
    if (std::string(name) == "small_batch_span") {  // developer knob (results identical for every value)
        if (value < 1) return fail(RQ_ERR_INVALID, "small_batch_span must be >= 1");
        g_sb_span = value;
        return RQ_OK;
    }
    if (std::string(name) == "small_batch") {  // 0 = batches of <= 64 queries take the few-launch path when it applies, 1 = never
        if (value < 0 || value > 1) return fail(RQ_ERR_INVALID, "small_batch must be 0 or 1");
        g_small_batch = value;
        return RQ_OK;
    }
    if (std::string(name) == "dense_dir") {  // test hook: 0 = run descriptors always appended and sorted, 1 = dense directories where they fit
        if (value < 0 || value > 1) return fail(RQ_ERR_INVALID, "dense_dir must be 0 or 1");
        g_dense_dir = value;
        return RQ_OK;
    }
    if (std::string(name) == "shared_thresholds") {  // rq_query_batch_sharded_device: 0 = every shard on its own thresholds, 1 = shared when world > 1, 2 = always
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "shared_thresholds must be 0, 1 or 2");
        g_shared_thr = value;
        return RQ_OK;
    }
    if (std::string(name) == "group_rank") {  // test hook: how a cluster-major stage places its pairs (0 atomics per pair, 1 auto, 2 ranked)
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "group_rank must be 0, 1 or 2");
        g_group_rank = value;
        return RQ_OK;
    }
    if (std::string(name) == "rerank_shadow") {  // fp16 shadow rows (rerank pre-filter) for indexes built / loaded from now on
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "rerank_shadow must be 0 (never), 1 (fp16 rows when they fit) or 2 (8-bit rows when they fit)");
        g_rerank_shadow = value;
        return RQ_OK;
    }
    if (std::string(name) == "scan_gate") {  // gate of the matrix-core scan (results never depend on it): 0 auto, 1 bf16 threshold, 2 additive where it exists
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "scan_gate must be 0, 1 or 2");
        g_scan_gate = value;
        return RQ_OK;
    }
    if (std::string(name) == "scan_debug") {
        // Measurement hooks that leave every result unchanged: 128 (kept for older hosts: the step counters are always on now),
        // 512 (rerank without the fp16 shadow rows), 4096 (phase stamps of the small-batch block), 16384 (stage list on stderr).
        // The timing ablations (1, 2, 4, 64, 1024, 8192: results are WRONG) and the in-kernel cycle counters (256) exist in
        // the developer build only (make dev -> librabitq_hip_dev.so, -DRQ_DEV_ABLATIONS).
#ifdef RQ_DEV_ABLATIONS
        const int allowed = 0x7FFFFFFF;
#else
        const int allowed = 128 | 512 | 4096 | 16384;
#endif
        if (value < 0 || (value & ~allowed)) return fail(RQ_ERR_INVALID, "scan_debug: this bit exists in the developer build only (librabitq_hip_dev.so)");
        g_scan_dbg = value;
        return RQ_OK;
    }
    return fail(RQ_ERR_INVALID, std::string("unknown option ") + name);
}

rq_status rq_set_profiling(int level) {
    if (level < 0 || level > 2) return fail(RQ_ERR_INVALID, "profiling level must be 0, 1 or 2");
    g_profiling = level;
    return RQ_OK;
}
rq_status rq_last_profile(rq_profile_t *out) {
    if (!out) return fail(RQ_ERR_INVALID, "null argument");
    return copy_out_sized(out, g_profile);
}

// ---- per-stage entry points --------------------------------------------------------------------
rq_status rq_rotate(const float *x, uint64_t n, uint32_t dim, const float *orthogonal, int use_mfma, float *out) {
    RQC(ensure_device());
    if (!x || !orthogonal || !out) return fail(RQ_ERR_INVALID, "null argument");
    if (dim == 0 || dim % 64) return fail(RQ_ERR_DIM_MISMATCH, "dim must be a multiple of 64");
    DevBuf<float> dx, dp, dout;
    RQC(dx.alloc(n * dim));
    RQC(dp.alloc((size_t)dim * dim));
    RQC(dout.alloc(n * dim));
    HIPC(hipMemcpy(dx.p, x, n * dim * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(dp.p, orthogonal, (size_t)dim * dim * 4, hipMemcpyHostToDevice));
    launch_rotate(dx.p, dp.p, dout.p, n, dim, use_mfma != 0, nullptr);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    HIPC(hipMemcpy(out, dout.p, n * dim * 4, hipMemcpyDeviceToHost));
    return RQ_OK;
}

rq_status rq_rotate_device(const rq_index *idx, const float *d_x, uint64_t n, float *d_out, float *out_ms) {
    RQC(ensure_device());
    if (!idx || !d_x || !d_out) return fail(RQ_ERR_INVALID, "null argument");
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    HIPC(hipEventRecord(e0, nullptr));
    launch_rotate(d_x, idx->P.p, d_out, n, idx->dim, true, nullptr);
    HIPC(hipEventRecord(e1, nullptr));
    HIPC(hipEventSynchronize(e1));
    HIPC(hipGetLastError());
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    if (out_ms) *out_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return RQ_OK;
}

rq_status rq_quantize_pack(const float *x_rot, uint64_t n, uint32_t dim, const float *centroids_rot, uint32_t k,
                           uint32_t *out_label, float *out_dist, uint64_t *out_codes, rq_factor_t *out_factors) {
    RQC(ensure_device());
    RQC(ensure_kernel_attributes());
    if (!x_rot || !centroids_rot || !out_label || !out_dist || !out_codes || !out_factors)
        return fail(RQ_ERR_INVALID, "null argument");
    if (dim == 0 || dim % 64 || k == 0) return fail(RQ_ERR_DIM_MISMATCH, "dim must be a multiple of 64, k > 0");
    rq_index tmp;
    tmp.dim = dim, tmp.k = k, tmp.W = dim / 64;
    DevBuf<float> dx, dd;
    DevBuf<uint32_t> dl;
    DevBuf<uint64_t> dc;
    DevBuf<float4> df;
    RQC(dx.alloc(n * dim));
    RQC(tmp.centroids.alloc((size_t)k * dim));
    RQC(tmp.cent_t.alloc((size_t)k * dim));
    RQC(dd.alloc(n));
    RQC(dl.alloc(n));
    RQC(dc.alloc(n * tmp.W));
    RQC(df.alloc(n));
    HIPC(hipMemcpy(dx.p, x_rot, n * dim * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(tmp.centroids.p, centroids_rot, (size_t)k * dim * 4, hipMemcpyHostToDevice));
    transpose_kernel<<<dim3(ceil_div(dim, 32), ceil_div(k, 32)), dim3(32, 8)>>>(tmp.centroids.p, tmp.cent_t.p, k, dim);
    AssignAux ax;  // the build's assignment path: matrix-core pre-filter + exact refinement (option assign_impl)
    HIPC(hipDeviceSynchronize());
    if (assign_has_mfma(tmp.W) && g_assign_impl.load() != 1) RQC(assign_aux_init(&tmp, ax, std::min<uint64_t>(std::max<uint64_t>(n, 1), 1ull << 20)));
    for (uint64_t i0 = 0; i0 < n; i0 += (1ull << 20)) {  // chunked: launches stay far below 2^32 threads
        const uint64_t m = std::min<uint64_t>(1ull << 20, n - i0);
        RQC(launch_assign_prefiltered(dx.p + i0 * dim, &tmp, ax, m, dl.p + i0, dd.p + i0));
        quantize_kernel<<<ceil_div(m, 32), 256>>>(dx.p + i0 * dim, tmp.centroids.p, dl.p + i0, m, dim,
                                                  dc.p + i0 * tmp.W, df.p + i0);
    }
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    if (n) {
        HIPC(hipMemcpy(out_label, dl.p, n * 4, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(out_dist, dd.p, n * 4, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(out_codes, dc.p, n * tmp.W * 8, hipMemcpyDeviceToHost));
        HIPC(hipMemcpy(out_factors, df.p, n * 16, hipMemcpyDeviceToHost));
    }
    return RQ_OK;
}

rq_status rq_coarse_rank(const rq_index *idx, const float *queries, uint32_t nq, uint32_t len, uint32_t probe,
                         float *out_y, uint32_t *out_cluster, float *out_dist) {
    RQC(ensure_device());
    if (!idx || !queries || !out_cluster || !out_dist) return fail(RQ_ERR_INVALID, "null argument");
    if (idx->dim != (len + 63) / 64 * 64) return fail(RQ_ERR_DIM_MISMATCH, "query length does not pad to dim");
    if (probe == 0) return fail(RQ_ERR_INVALID, "probe == 0");
    const uint32_t dim = idx->dim, k = idx->k, nprobe = std::min(probe, k);
    if (nprobe > RQ_MAX_PROBE) return fail(RQ_ERR_UNSUPPORTED, "probe > 16384");
    DevBuf<float> dq, qpad, y, dist, pd;
    DevBuf<uint32_t> pc;
    RQC(dq.alloc((uint64_t)nq * len));
    RQC(qpad.alloc((uint64_t)nq * dim));
    RQC(y.alloc((uint64_t)nq * dim));
    RQC(dist.alloc((uint64_t)nq * k));
    RQC(pd.alloc((uint64_t)nq * nprobe));
    RQC(pc.alloc((uint64_t)nq * nprobe));
    HIPC(hipMemcpy(dq.p, queries, (uint64_t)nq * len * 4, hipMemcpyHostToDevice));
    pad_rows_kernel<<<ceil_div((uint64_t)nq * dim, 256), 256>>>(dq.p, qpad.p, nq, len, dim);
    launch_rotate(qpad.p, idx->P.p, y.p, nq, dim, nq >= 32, nullptr);
    coarse_dist_kernel<4><<<dim3(ceil_div(nq, 4), ceil_div(k, 256)), 256, 4 * dim * sizeof(float)>>>(
        idx->cent_t.p, y.p, dist.p, k, dim, nq, k);
    launch_select(dist.p, k, nprobe, pc.p, pd.p, 0, nprobe, nq, nullptr);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    if (out_y) HIPC(hipMemcpy(out_y, y.p, (uint64_t)nq * dim * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(out_cluster, pc.p, (uint64_t)nq * nprobe * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(out_dist, pd.p, (uint64_t)nq * nprobe * 4, hipMemcpyDeviceToHost));
    return RQ_OK;
}

rq_status rq_query_prep(const rq_index *idx, const float *y, uint32_t nq, const uint32_t *cluster, float *out_lower,
                        float *out_delta, uint32_t *out_sum, uint64_t *out_planes) {
    RQC(ensure_device());
    if (!idx || !y || !cluster || !out_lower || !out_delta || !out_sum || !out_planes)
        return fail(RQ_ERR_INVALID, "null argument");
    const uint32_t dim = idx->dim, W = idx->W;
    for (uint32_t i = 0; i < nq; ++i)
        if (cluster[i] >= idx->k) return fail(RQ_ERR_INVALID, "cluster id out of range");
    DevBuf<float> dy, ycd;
    DevBuf<uint32_t> dc, dsum;
    DevBuf<PairScalars> scal;
    DevBuf<uint64_t> planes;
    RQC(dy.alloc((uint64_t)nq * dim));
    RQC(ycd.alloc(nq));
    RQC(dc.alloc(nq));
    RQC(dsum.alloc(nq));
    RQC(scal.alloc(nq));
    RQC(planes.alloc((uint64_t)nq * 4 * W));
    HIPC(hipMemcpy(dy.p, y, (uint64_t)nq * dim * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(dc.p, cluster, nq * 4, hipMemcpyHostToDevice));
    HIPC(hipMemset(ycd.p, 0, nq * 4));
    prep_kernel<<<ceil_div(nq, 4), 256>>>(dy.p, idx->centroids.p, idx->offsets.p, dc.p, ycd.p, nq, 1, dim, scal.p,
                                          planes.p, nullptr, nullptr, dsum.p, idx->k, 0u);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    std::vector<PairScalars> hs(nq);
    HIPC(hipMemcpy(hs.data(), scal.p, nq * sizeof(PairScalars), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < nq; ++i) out_lower[i] = hs[i].lower, out_delta[i] = hs[i].delta;
    HIPC(hipMemcpy(out_sum, dsum.p, nq * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(out_planes, planes.p, (uint64_t)nq * 4 * W * 8, hipMemcpyDeviceToHost));
    return RQ_OK;
}

rq_status rq_scan(const rq_index *idx, uint32_t cluster, float y_c_distance_square, const uint64_t *planes,
                  float lower_bound, float scalar_sum, float delta, float *out_rough) {
    RQC(ensure_device());
    if (!idx || !planes || !out_rough) return fail(RQ_ERR_INVALID, "null argument");
    if (cluster >= idx->k) return fail(RQ_ERR_INVALID, "cluster id out of range");
    uint32_t off[2];
    HIPC(hipMemcpy(off, idx->offsets.p + cluster, 8, hipMemcpyDeviceToHost));
    const uint32_t len = off[1] - off[0];
    if (len == 0) return RQ_OK;
    DevBuf<uint64_t> dpl;
    DevBuf<float> dout;
    RQC(dpl.alloc(4 * idx->W));
    RQC(dout.alloc(len));
    HIPC(hipMemcpy(dpl.p, planes, 4 * idx->W * 8, hipMemcpyHostToDevice));
    scan_dense_kernel<<<ceil_div(len, 256), 256>>>(reinterpret_cast<const uint32_t *>(idx->codes.p), idx->factors.p,
                                                   off[0], len, idx->W, reinterpret_cast<const uint32_t *>(dpl.p),
                                                   lower_bound, delta, scalar_sum, y_c_distance_square, dout.p);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    HIPC(hipMemcpy(out_rough, dout.p, len * 4, hipMemcpyDeviceToHost));
    return RQ_OK;
}

rq_status rq_rerank(const rq_index *idx, const float *query_padded, const uint32_t *pos, uint32_t m,
                    float *out_accurate) {
    RQC(ensure_device());
    if (!idx || !query_padded || !pos || !out_accurate) return fail(RQ_ERR_INVALID, "null argument");
    for (uint32_t i = 0; i < m; ++i)
        if (pos[i] >= idx->n) return fail(RQ_ERR_INVALID, "position out of range");
    if (m == 0) return RQ_OK;
    DevBuf<float> dq, dout;
    DevBuf<uint32_t> dp;
    RQC(dq.alloc(idx->dim));
    RQC(dout.alloc(m));
    RQC(dp.alloc(m));
    HIPC(hipMemcpy(dq.p, query_padded, idx->dim * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(dp.p, pos, m * 4, hipMemcpyHostToDevice));
    accurate_flat_kernel<<<ceil_div(m, 32), 256>>>(dp.p, m, idx->view(), dq.p, idx->dim, dout.p);
    HIPC(hipDeviceSynchronize());
    HIPC(hipGetLastError());
    HIPC(hipMemcpy(out_accurate, dout.p, m * 4, hipMemcpyDeviceToHost));
    return RQ_OK;
}

}  // extern "C"
