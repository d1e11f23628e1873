/*
 * rabitq_oracle.h -- CPU oracle for the RaBitQ build/query hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rabitq_amd/ may include, link, import or
 * execute anything in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED BY THE REFERENCE: kemingy/rabitq ships no tests, golden vectors or
 * known-answer fixtures for this path, and its Rust toolchain is not
 * available here, so the reference itself cannot be run.  This restatement is pinned by
 *   (1) hand-derived known-answer tests read off the reference source (tests/test_oracle_kat.py),
 *   (2) an independent numpy restatement of the reference's scalar `*_raw` fallbacks
 *       (oracle/raw_numpy.py) that must agree exactly on all integer work,
 *   (3) structural invariants of the index (tests/test_oracle_index.py).
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * The parity target is the reference's x86 AVX2 path (runtime-dispatched in src/utils.rs).
 */
#ifndef RABITQ_ORACLE_H
#define RABITQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- src/consts.rs:4-12 ---- */
#define RQO_DEFAULT_X_DOT_PRODUCT 0.8f
#define RQO_EPSILON 1.9f
#define RQO_THETA_LOG_DIM 4
#define RQO_WINDOW_SIZE 12

/* ---- src/simd.rs kernels (AVX2/FMA restatements, same lane and reduction order) ---- */
float rqo_l2_squared_distance(const float *lhs, const float *rhs, size_t n);      /* simd.rs:14-73   */
float rqo_vector_dot_product(const float *lhs, const float *rhs, size_t n);       /* simd.rs:257-314 */
void rqo_min_max_residual(float *res, const float *x, const float *y, size_t n,
                          float *out_min, float *out_max);                        /* simd.rs:117-173 */
uint32_t rqo_scalar_quantize(uint8_t *quantized, const float *vec, size_t n, float lower_bound,
                             float multiplier);                                   /* simd.rs:185-247 */
void rqo_vector_binarize_query(const uint8_t *vec, size_t n, uint64_t *binary);   /* simd.rs:83-107  */
uint32_t rqo_binary_dot_product(const uint64_t *lhs, const uint64_t *rhs, size_t nwords); /* simd.rs:326-384 */

/* ---- src/utils.rs ---- */
uint32_t rqo_asymmetric_binary_dot_product(const uint64_t *x, const uint64_t *y,
                                           size_t nwords);                         /* utils.rs:113-135 */
void rqo_vector_binarize_u64(const float *vec, size_t n, uint64_t *binary);        /* utils.rs:53-61   */
void rqo_project(const float *vec, const float *orthogonal_t, size_t dim, float *out); /* utils.rs:237-258 */
void rqo_kmeans_nearest_cluster(const float *centroids, size_t k, size_t dim, const float *vec,
                                uint32_t *out_label, float *out_dist);             /* utils.rs:261-277 */
float rqo_calculate_recall(const int32_t *truth, size_t ntruth, const int32_t *res,
                           size_t topk);                                           /* utils.rs:367-379 */

/* ---- src/ord32.rs:12-26 ---- */
int32_t rqo_ord32_from_f32(float x);
float rqo_ord32_to_f32(int32_t key);

/* ---- src/metrics.rs:7-65 ---- */
typedef struct {
    uint64_t rough, precise, query, miss;
} rqo_metrics_t;
void rqo_metrics_get(rqo_metrics_t *out);
void rqo_metrics_reset(void);

/* ---- src/rabitq.rs:21-32 ---- */
typedef struct {
    float factor_ip, factor_ppc, error_bound, center_distance_square;
} rqo_factor_t;

/* ---- src/rabitq.rs:57-68, arrays owned by the index ---- */
typedef struct {
    uint32_t dim;          /* padded, multiple of 64 */
    uint64_t n;
    uint32_t k;
    float *base;           /* n x dim, row i = vector at cluster-order position i (un-rotated)  */
    float *orthogonal;     /* dim x dim row-major: orthogonal[r*dim + c] = P(r, c)                */
    float *orthogonal_t;   /* transpose of the above (columns of P contiguous, as faer stores it) */
    float *centroids;      /* k x dim, row j = rotated centroid j (faer: dim x k col-major)       */
    uint32_t *offsets;     /* k + 1 */
    uint32_t *map_ids;     /* n : cluster-order position -> original id */
    uint64_t *x_binary_vec;/* n * dim/64 */
    rqo_factor_t *factors; /* n */
} rqo_index_t;

/* RaBitQ::from_path on in-memory arrays: src/rabitq.rs:159-265.  `orthogonal` (d_pad x d_pad,
 * row-major) is an INPUT because the reference draws it from an unseeded RNG (utils.rs:16-20). */
rqo_index_t *rqo_build(const float *base, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                       const float *orthogonal);
/* same, but from already-rotated vectors/centroids (skips rabitq.rs:188-189); used to test the
 * GPU quantize+pack kernel on identical inputs. base_orig may be NULL (then base = zeros). */
void rqo_free(rqo_index_t *idx);
/* src/rabitq.rs:84-156 (five "vecs" files; byte layout in src/utils.rs:280-364) */
int rqo_dump_dir(const rqo_index_t *idx, const char *dir);
rqo_index_t *rqo_load_dir(const char *dir);

/* RaBitQ::query: src/rabitq.rs:268-333.  Returns 0 on success, <0 where the reference panics.
 * out_dist/out_id need room for topk entries (heap ranker) and are filled in the heap's internal
 * (Rust BinaryHeap Vec) order; heuristic ranker results are sorted ascending. */
int rqo_query(const rqo_index_t *idx, const float *query, uint32_t len, uint32_t probe,
              uint32_t topk, int heuristic_rank, float *out_dist, uint32_t *out_id, uint32_t *out_n);

/* Query-side stage outputs for one (query, cluster) pair, for per-kernel parity tests. */
/* coarse ranking: rabitq.rs:283-297. out arrays have room for min(probe,k). */
int rqo_coarse_rank(const rqo_index_t *idx, const float *y_projected, uint32_t probe,
                    uint32_t *out_cluster, float *out_dist);
/* per-cluster query quantisation: rabitq.rs:304-317 */
void rqo_query_prep(const rqo_index_t *idx, const float *y_projected, uint32_t cluster,
                    float *out_lower, float *out_delta, uint32_t *out_sum, uint64_t *out_planes);
/* calculate_rough_distance for the whole list of `cluster`: rabitq.rs:336-367.
 * out_rough needs offsets[c+1]-offsets[c] floats. */
void rqo_scan_cluster(const rqo_index_t *idx, uint32_t cluster, float y_c_distance_square,
                      const uint64_t *planes, float lower_bound, float scalar_sum, float delta,
                      float *out_rough);

/* Scan-only timing helper for the CPU baseline: runs rotate + coarse rank + prep + scan of every
 * probed list (no rerank) and returns the number of candidates scanned. */
uint64_t rqo_scan_only(const rqo_index_t *idx, const float *query, uint32_t len, uint32_t probe,
                       float *scratch_rough);

/* Wrap caller-owned arrays as an index without copying (the CPU baseline at >= 100M borrows the
 * arrays the GPU engine built).  Free with rqo_free_view (frees only the struct + P transpose). */
rqo_index_t *rqo_view(uint32_t dim, uint64_t n, uint32_t k, const float *base, const float *orthogonal,
                      const float *centroids, const uint32_t *offsets, const uint32_t *map_ids,
                      const uint64_t *codes, const float *factors);
void rqo_free_view(rqo_index_t *idx);

#ifdef __cplusplus
}
#endif
#endif
