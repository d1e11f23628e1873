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
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <dlfcn.h>
#include <queue>
#include <map>

#include "common.h"
#include "kernels_build.h"
#include "kernels_query.h"
#include "kernels_small.h"

#include "host_state.h"
#include "host_query.h"
#include "host_build.h"

// ------------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------------
extern "C" {

const char *rq_version(void) { return "rabitq_hip 0.5.0 (gfx950, abi 4)"; }
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
    full.split_rows = idx->split_rows ? 1u : 0u;
    return copy_out_sized(out, full);
}

rq_status rq_get_device_ptr(const rq_index *idx, int which, const void **out_ptr, uint64_t *out_bytes) {
    if (!idx || !out_ptr || !out_bytes) return fail(RQ_ERR_INVALID, "null argument");
    switch (which) {
        case RQ_ARR_BASE:
            if (idx->n_dev < idx->n)  // tiered: the HBM tier holds packed list heads, not rows at their positions
                return fail(RQ_ERR_UNSUPPORTED, "the raw vectors of this index are tiered (HBM + pinned host memory): no single device array; use rq_get_array");
            if (idx->split_rows)  // (common.h) not an array of f32 rows
                return fail(RQ_ERR_UNSUPPORTED, "the raw vectors of this index are stored as split rows (two 16-bit planes per row): no f32 device array; use rq_get_array");
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
    if (idx && which == RQ_ARR_BASE && (idx->n_dev < idx->n || idx->split_rows)) {  // both tiers / split rows
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

#include "host_shard.h"

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
    if (std::string(name) == "stage_growth") {  // geometric growth of the early stages (0 = default: 16 below 32 768 queries, 8 from there on)
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
    if (std::string(name) == "pass_overlap") {  // a call of several passes: two in flight (1, default) or one after the other (0); results identical
        if (value < 0 || value > 1) return fail(RQ_ERR_INVALID, "pass_overlap must be 0 or 1");
        g_pass_overlap = value;
        return RQ_OK;
    }
    if (std::string(name) == "split_rows") {  // tiered indexes built / loaded from now on: raw vectors as two 16-bit planes (1, default) or plain f32 (0)
        if (value < 0 || value > 2) return fail(RQ_ERR_INVALID, "split_rows must be 0, 1 or 2");
        g_split_rows = value;
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
