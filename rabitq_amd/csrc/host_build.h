// host_build.h -- part of the host side of librabitq_hip.so (one translation unit: rabitq_hip.hip includes the host_*.h files in order;
// they are not stand-alone headers).  Index construction (RaBitQ::from_path, src/rabitq.rs:159-265): derived state, rotation matrix, assignment, HBM / host tiers, the streamed builder; vecs I/O and the JSON image (src/rabitq.rs:72-156).
#pragma once
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
    if (!kind || idx->base_host != nullptr || idx->split_rows || idx->n < RQ_SHADOW_MIN_ROWS) return RQ_OK;
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
static std::atomic<int> g_split_rows{1};           // indexes built from now on with no room for shadow rows keep their raw vectors as split rows (0: plain f32, the round-4 layout; 2: always)
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
        // Will shadow rows fit beside them (derive_shadow_rows' rule, asked before the rows are placed)?  If not, the rows are stored
        // as split rows (common.h) and their own first plane is the re-ranker's pre-filter.  (2: always -- test hook.)
        const int sr = g_split_rows.load(), kind = g_rerank_shadow.load();
        if (sr == 2) {
            idx->split_rows = true;
        } else if (sr == 1 && kind != 0) {
            size_t free_b = 0, total_b = 0;
            HIPC(hipMemGetInfo(&free_b, &total_b));
            const uint64_t shadow = idx->n * idx->dim * (kind == 2 && idx->dim <= 4096 ? 1ull : 2ull);
            idx->split_rows = shadow > (1ull << 30) && free_b < want + shadow + (48ull << 30);
        }
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
    idx->split_rows = g_split_rows.load() != 0;  // both tiers as split rows (common.h): the re-ranker's pre-filter where no shadow fits
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

// Split rows <-> plain f32 rows on the host (dump / load / rq_get_array of an index that stores split rows: common.h), rows shared
// among up to 16 threads: a 100M x 768 index is 7.7e10 words
static void transcode_split_rows(float *plain, uint32_t *planes, uint64_t rows, uint32_t dim, bool to_planes) {
    auto work = [=](uint64_t r0, uint64_t r1) {
        for (uint64_t r = r0; r < r1; ++r) {
            float *pr = plain + r * dim;
            uint16_t *h = reinterpret_cast<uint16_t *>(planes + r * dim), *lo = h + dim;
            if (to_planes) {
                for (uint32_t e = 0; e < dim; ++e) {
                    uint32_t bits;
                    memcpy(&bits, pr + e, 4);
                    h[e] = (uint16_t)((bits + 0x8000u) >> 16), lo[e] = (uint16_t)bits;
                }
            } else {
                for (uint32_t e = 0; e < dim; ++e) {
                    const uint32_t bits = rq_split_join(h[e], lo[e]);
                    memcpy(pr + e, &bits, 4);
                }
            }
        }
    };
    const uint64_t words = rows * dim;
    const unsigned nt = words < (1u << 22) ? 1u : std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (nt <= 1) return work(0, rows);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, rows * t / nt, rows * (t + 1) / nt);
    for (auto &x : th) x.join();
}

// rows [i0, i0 + m) of the cluster-ordered base between host memory (`buf`, m x dim) and whichever tier holds them
// (to_index = false: index -> buf; true: buf -> index)
static rq_status copy_base_rows(const rq_index *idx, uint64_t i0, uint64_t m, float *buf, bool to_index) {
    const uint64_t dim = idx->dim, i1 = i0 + m;
    std::vector<uint32_t> planes;  // split rows (common.h): transcoded here, on the host -- this is the dump / load path
    auto dev_copy = [&](uint64_t dev_row, uint64_t rows, float *hp) -> rq_status {
        if (!rows) return RQ_OK;
        if (!idx->split_rows) {
            if (to_index) HIPC(hipMemcpy(idx->base.p + dev_row * dim, hp, rows * dim * 4, hipMemcpyHostToDevice));
            else HIPC(hipMemcpy(hp, idx->base.p + dev_row * dim, rows * dim * 4, hipMemcpyDeviceToHost));
            return RQ_OK;
        }
        // (in pieces of at most 256 MiB: the staging buffer must not double a 280 GB tier)
        const uint64_t piece = std::max<uint64_t>(1, (64ull << 20) / dim);
        for (uint64_t r0 = 0; r0 < rows; r0 += piece) {
            const uint64_t nr = std::min(piece, rows - r0);
            planes.resize(nr * dim);
            if (to_index) {
                transcode_split_rows(hp + r0 * dim, planes.data(), nr, (uint32_t)dim, true);
                HIPC(hipMemcpy(idx->base.p + (dev_row + r0) * dim, planes.data(), nr * dim * 4, hipMemcpyHostToDevice));
            } else {
                HIPC(hipMemcpy(planes.data(), idx->base.p + (dev_row + r0) * dim, nr * dim * 4, hipMemcpyDeviceToHost));
                transcode_split_rows(hp + r0 * dim, planes.data(), nr, (uint32_t)dim, false);
            }
        }
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
            float *brow = buf + (a2 - i0) * dim;
            if (!idx->split_rows) {
                if (to_index) memcpy(hrow, brow, (e - a2) * dim * 4);
                else memcpy(brow, hrow, (e - a2) * dim * 4);
            } else {
                transcode_split_rows(brow, reinterpret_cast<uint32_t *>(hrow), e - a2, (uint32_t)dim, to_index);
            }
        }
    }
    return RQ_OK;
}

