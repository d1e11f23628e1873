/*
 * rabitq_oracle.c -- CPU restatement (C11 + AVX2/FMA intrinsics) of kemingy/rabitq's
 * build/query hot path.  TEST INFRASTRUCTURE ONLY -- see rabitq_oracle.h for the rules and for
 * the "parity unpinned by the reference" statement.
 *
 * Compile with:  gcc -O2 -mavx2 -mfma -ffp-contract=off -fno-fast-math   (oracle/Makefile)
 * `-ffp-contract=off` matters: rustc never contracts a*b+c, so every fused multiply-add below is
 * an explicit intrinsic exactly where the reference uses `_mm256_fmadd_ps`.
 *
 * Third-party arithmetic the reference delegates to faer 0.19.4 (source not in /root/reference):
 *   - rabitq.rs:188-189  base*P, centroids*P (GEMM)      -> restated in the order of the reference's
 *                                                           own query-side `project` (utils.rs:237-258)
 *   - rabitq.rs:206      norm_l2 of the residual         -> sqrtf of the AVX2-ordered sum of squares
 *   - rabitq.rs:212      <residual, sign>                -> AVX2 `vector_dot_product` order
 * No reference test pins those call sites; results agree with faer to f32 rounding (stated
 * tolerance rtol 1e-5 in tests), and everything downstream is checked on identical inputs.
 */
#include "rabitq_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <float.h>

/* ------------------------------------------------------------------------------------------ */
/* src/simd.rs                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* simd.rs:52-63: fold the 8 lanes as ((a0+a4)+(a1+a5)) + ((a2+a6)+(a3+a7)). */
static inline float reduce8(__m256 acc) {
    __m256 swapped = _mm256_permute2f128_ps(acc, acc, 1);
    __m256 c = _mm256_add_ps(acc, swapped);
    c = _mm256_hadd_ps(c, c);
    c = _mm256_hadd_ps(c, c);
    return _mm256_cvtss_f32(c);
}

/* simd.rs:14-73.  One 8-lane accumulator; per 8-chunk: diff = a-b (rounded), sum = fma(diff,diff,sum)
 * in index order (the 2x-unrolled loop and the 8-remainder loop are the same chain); scalar tail is
 * an un-fused multiply-add. */
float rqo_l2_squared_distance(const float *lhs, const float *rhs, size_t n) {
    __m256 sum = _mm256_setzero_ps();
    size_t chunks = n / 8;
    for (size_t c = 0; c < chunks; ++c) {
        __m256 d = _mm256_sub_ps(_mm256_loadu_ps(lhs + 8 * c), _mm256_loadu_ps(rhs + 8 * c));
        sum = _mm256_fmadd_ps(d, d, sum);
    }
    float res = reduce8(sum);
    for (size_t i = chunks * 8; i < n; ++i) {
        float r = lhs[i] - rhs[i];
        res += r * r;
    }
    return res;
}

/* simd.rs:257-314.  Same structure with acc = fma(x, y, acc). */
float rqo_vector_dot_product(const float *lhs, const float *rhs, size_t n) {
    __m256 acc = _mm256_setzero_ps();
    size_t chunks = n / 8;
    for (size_t c = 0; c < chunks; ++c)
        acc = _mm256_fmadd_ps(_mm256_loadu_ps(lhs + 8 * c), _mm256_loadu_ps(rhs + 8 * c), acc);
    float sum = reduce8(acc);
    for (size_t i = chunks * 8; i < n; ++i) sum += lhs[i] * rhs[i];
    return sum;
}

/* simd.rs:117-173.  res = x - y; lane-wise min/max from +/-f32::MAX, then strict </> folds. */
void rqo_min_max_residual(float *res, const float *x, const float *y, size_t n, float *out_min,
                          float *out_max) {
    __m256 vmin = _mm256_set1_ps(FLT_MAX), vmax = _mm256_set1_ps(-FLT_MAX);
    size_t chunks = n / 8;
    for (size_t c = 0; c < chunks; ++c) {
        __m256 r = _mm256_sub_ps(_mm256_loadu_ps(x + 8 * c), _mm256_loadu_ps(y + 8 * c));
        _mm256_storeu_ps(res + 8 * c, r);
        vmin = _mm256_min_ps(vmin, r);
        vmax = _mm256_max_ps(vmax, r);
    }
    float lanes[8], mn = FLT_MAX, mx = -FLT_MAX;
    _mm256_storeu_ps(lanes, vmin);
    for (int i = 0; i < 8; ++i)
        if (lanes[i] < mn) mn = lanes[i];
    _mm256_storeu_ps(lanes, vmax);
    for (int i = 0; i < 8; ++i)
        if (lanes[i] > mx) mx = lanes[i];
    for (size_t i = chunks * 8; i < n; ++i) {
        res[i] = x[i] - y[i];
        if (res[i] < mn) mn = res[i];
        if (res[i] > mx) mx = res[i];
    }
    *out_min = mn;
    *out_max = mx;
}

/* simd.rs:185-247.  q = cvtps_epi32((v - lo) * mult): subtract then multiply (not fused), MXCSR
 * round-to-nearest-even, no bias (simd.rs:177).  The low byte of every i32 is stored; the i32 lanes
 * are summed with wrap-around.  Tail (n % 8, never hit when dim % 64 == 0) rounds half away from
 * zero (`f32::round`). */
uint32_t rqo_scalar_quantize(uint8_t *quantized, const float *vec, size_t n, float lower_bound,
                             float multiplier) {
    const __m256 lo = _mm256_set1_ps(lower_bound), mult = _mm256_set1_ps(multiplier);
    __m256i sum = _mm256_setzero_si256();
    size_t chunks = n / 8;
    for (size_t c = 0; c < chunks; ++c) {
        __m256 v = _mm256_loadu_ps(vec + 8 * c);
        __m256i q = _mm256_cvtps_epi32(_mm256_mul_ps(_mm256_sub_ps(v, lo), mult));
        sum = _mm256_add_epi32(sum, q);
        int32_t lanes[8];
        _mm256_storeu_si256((__m256i *)lanes, q);
        for (int i = 0; i < 8; ++i) quantized[8 * c + i] = (uint8_t)(lanes[i] & 0xff);
    }
    int32_t s[8];
    _mm256_storeu_si256((__m256i *)s, sum);
    uint32_t total = 0;
    for (int i = 0; i < 8; ++i) total += (uint32_t)s[i];
    for (size_t i = chunks * 8; i < n; ++i) {
        uint8_t q = (uint8_t)roundf((vec[i] - lower_bound) * multiplier);
        quantized[i] = q;
        total += q;
    }
    return total;
}

/* simd.rs:83-107.  Plane p (bit p of each 4-bit code) occupies words [p*n/64, (p+1)*n/64); inside a
 * word, bit b <-> dimension 64*w + b.  Caller zeroes `binary` (rabitq.rs:316). */
void rqo_vector_binarize_query(const uint8_t *vec, size_t n, uint64_t *binary) {
    size_t words = n >> 6;
    for (size_t i = 0; i < n; i += 32) {
        __m256i v = _mm256_loadu_si256((const __m256i *)(vec + i));
        v = _mm256_slli_epi32(v, 4); /* bit 3 of every byte -> its MSB */
        for (int j = 0; j < RQO_THETA_LOG_DIM; ++j) {
            uint64_t mask = (uint64_t)(uint32_t)_mm256_movemask_epi8(v);
            binary[(size_t)(3 - j) * words + (i >> 6)] |= mask << (i & 32);
            v = _mm256_slli_epi32(v, 1);
        }
    }
}

/* simd.rs:346-361: per-64-bit-lane popcount by nibble lookup + psadbw. */
static inline __m256i popcnt_epi64(__m256i x) {
    const __m256i lut = _mm256_setr_epi8(0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4, 0, 1, 1, 2, 1,
                                         2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4);
    const __m256i nib = _mm256_set1_epi8(15);
    __m256i lo = _mm256_shuffle_epi8(lut, _mm256_and_si256(x, nib));
    __m256i hi = _mm256_shuffle_epi8(lut, _mm256_and_si256(_mm256_srli_epi64(x, 4), nib));
    return _mm256_sad_epu8(_mm256_add_epi8(lo, hi), _mm256_setzero_si256());
}

/* simd.rs:326-384.  popcount(lhs & rhs).  Fewer than 4 words (D < 256, so D = 128 always) takes the
 * scalar count_ones branch (simd.rs:333-339) exactly as the reference does. */
uint32_t rqo_binary_dot_product(const uint64_t *lhs, const uint64_t *rhs, size_t nwords) {
    uint32_t sum = 0;
    size_t quads = nwords / 4;
    if (quads == 0) {
        for (size_t i = 0; i < nwords; ++i) sum += (uint32_t)__builtin_popcountll(lhs[i] & rhs[i]);
        return sum;
    }
    for (size_t i = 4 * quads; i < nwords; ++i)
        sum += (uint32_t)__builtin_popcountll(lhs[i] & rhs[i]);
    __m256i acc = _mm256_setzero_si256();
    for (size_t q = 0; q < quads; ++q) {
        __m256i a = _mm256_loadu_si256((const __m256i *)(lhs + 4 * q));
        __m256i b = _mm256_loadu_si256((const __m256i *)(rhs + 4 * q));
        acc = _mm256_add_epi64(acc, popcnt_epi64(_mm256_and_si256(a, b)));
    }
    uint64_t lanes[4];
    _mm256_storeu_si256((__m256i *)lanes, acc);
    sum += (uint32_t)(lanes[0] + lanes[1] + lanes[2] + lanes[3]);
    return sum;
}

/* ------------------------------------------------------------------------------------------ */
/* src/utils.rs                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* utils.rs:113-135: sum_p popcount(x & plane_p) << p. */
uint32_t rqo_asymmetric_binary_dot_product(const uint64_t *x, const uint64_t *y, size_t nwords) {
    uint32_t res = 0;
    for (int p = 0; p < RQO_THETA_LOG_DIM; ++p)
        res += rqo_binary_dot_product(x, y + (size_t)p * nwords, nwords) << p;
    return res;
}

/* utils.rs:53-61: bit set iff v > 0.0 strictly (zero, negatives and NaN -> 0). */
void rqo_vector_binarize_u64(const float *vec, size_t n, uint64_t *binary) {
    size_t words = (n + 63) / 64;
    memset(binary, 0, words * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i)
        if (vec[i] > 0.0f) binary[i / 64] |= (uint64_t)1 << (i % 64);
}

/* utils.rs:237-258 (AVX2 branch): out[i] = vector_dot_product(vec, column i of P).
 * orthogonal_t holds P transposed so that column i is contiguous (faer is column-major). */
void rqo_project(const float *vec, const float *orthogonal_t, size_t dim, float *out) {
    for (size_t i = 0; i < dim; ++i)
        out[i] = rqo_vector_dot_product(vec, orthogonal_t + i * dim, dim);
}

/* utils.rs:261-277: first minimum wins (strict <), starting from (label 0, f32::MAX). */
void rqo_kmeans_nearest_cluster(const float *centroids, size_t k, size_t dim, const float *vec,
                                uint32_t *out_label, float *out_dist) {
    float best = FLT_MAX;
    uint32_t label = 0;
    for (size_t j = 0; j < k; ++j) {
        float d = rqo_l2_squared_distance(centroids + j * dim, vec, dim);
        if (d < best) {
            best = d;
            label = (uint32_t)j;
        }
    }
    *out_label = label;
    *out_dist = best;
}

/* utils.rs:367-379 */
float rqo_calculate_recall(const int32_t *truth, size_t ntruth, const int32_t *res, size_t topk) {
    size_t lim = ntruth < topk ? ntruth : topk, count = 0;
    for (size_t i = 0; i < topk; ++i)
        for (size_t t = 0; t < lim; ++t)
            if (res[i] == truth[t]) {
                ++count;
                break;
            }
    return (float)count / (float)topk;
}

/* ------------------------------------------------------------------------------------------ */
/* src/ord32.rs:12-26                                                                         */
/* ------------------------------------------------------------------------------------------ */
int32_t rqo_ord32_from_f32(float x) {
    int32_t bits;
    memcpy(&bits, &x, 4);
    uint32_t mask = ((uint32_t)(bits >> 31)) >> 1; /* arithmetic shift, then logical */
    return bits ^ (int32_t)mask;
}
float rqo_ord32_to_f32(int32_t key) {
    uint32_t mask = ((uint32_t)(key >> 31)) >> 1;
    int32_t bits = key ^ (int32_t)mask;
    float x;
    memcpy(&x, &bits, 4);
    return x;
}

/* ------------------------------------------------------------------------------------------ */
/* src/metrics.rs                                                                             */
/* ------------------------------------------------------------------------------------------ */
static rqo_metrics_t g_metrics;
void rqo_metrics_get(rqo_metrics_t *out) { *out = g_metrics; }
void rqo_metrics_reset(void) { memset(&g_metrics, 0, sizeof g_metrics); }

/* ------------------------------------------------------------------------------------------ */
/* Rust std::collections::BinaryHeap<(Ord32, AlwaysEqual<u32>)> (max-heap on the Ord32 key only,   */
/* src/ord32.rs:43-66).  push = append + sift_up; pop = swap last into the root,                  */
/* sift_down_to_bottom, then sift_up (alloc::collections::binary_heap, Rust 1.7x).  Restated so    */
/* that which of several equal keys is evicted matches the reference.                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t key;
    uint32_t id;
} heap_item_t;

typedef struct {
    heap_item_t *data;
    size_t len, cap;
} heap_t;

static void heap_sift_up(heap_t *h, size_t start, size_t pos) {
    heap_item_t hole = h->data[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (hole.key <= h->data[parent].key) break;
        h->data[pos] = h->data[parent];
        pos = parent;
    }
    h->data[pos] = hole;
}

static void heap_push(heap_t *h, heap_item_t item) {
    if (h->len == h->cap) {
        h->cap = h->cap ? h->cap * 2 : 16;
        h->data = (heap_item_t *)realloc(h->data, h->cap * sizeof(heap_item_t));
    }
    h->data[h->len] = item;
    heap_sift_up(h, 0, h->len);
    h->len++;
}

static void heap_pop(heap_t *h) {
    if (h->len == 0) return;
    heap_item_t last = h->data[--h->len];
    if (h->len == 0) return;
    /* item <-> data[0]; then sift_down_to_bottom(0) */
    size_t end = h->len, pos = 0;
    heap_item_t hole = last;
    size_t child = 1;
    while (child + 1 < end) { /* child <= end.saturating_sub(2) */
        if (h->data[child].key <= h->data[child + 1].key) child += 1;
        h->data[pos] = h->data[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        h->data[pos] = h->data[child];
        pos = child;
    }
    h->data[pos] = hole;
    heap_sift_up(h, 0, pos);
}

/* ------------------------------------------------------------------------------------------ */
/* src/rerank.rs                                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int heuristic;
    float threshold;
    size_t topk;
    const float *query;
    /* heap ranker (rerank.rs:61-114) */
    heap_t heap;
    /* heuristic ranker (rerank.rs:117-177) */
    float recent_max_accurate;
    size_t count;
    float *arr_dist;
    uint32_t *arr_id;
    size_t arr_len, arr_cap;
} reranker_t;

static void reranker_init(reranker_t *r, const float *query, size_t topk, int heuristic) {
    memset(r, 0, sizeof *r);
    r->heuristic = heuristic;
    r->threshold = FLT_MAX;
    r->recent_max_accurate = -FLT_MAX; /* f32::MIN */
    r->topk = topk;
    r->query = query;
}

static void reranker_free(reranker_t *r) {
    free(r->heap.data);
    free(r->arr_dist);
    free(r->arr_id);
}

/* rerank.rs:81-106 (heap) and :143-168 (heuristic).  rough[i] pairs with position first + i. */
static void rank_batch(reranker_t *r, const float *rough, uint32_t first, uint32_t count,
                       const rqo_index_t *idx) {
    uint64_t precise = 0;
    for (uint32_t i = 0; i < count; ++i) {
        if (!(rough[i] < r->threshold)) continue;
        uint32_t u = first + i;
        float accurate = rqo_l2_squared_distance(idx->base + (size_t)u * idx->dim, r->query, idx->dim);
        ++precise;
        if (!(accurate < r->threshold)) continue;
        if (!r->heuristic) {
            heap_item_t it = {rqo_ord32_from_f32(accurate), idx->map_ids[u]};
            heap_push(&r->heap, it);
            if (r->heap.len > r->topk) heap_pop(&r->heap);
            if (r->heap.len == r->topk) r->threshold = rqo_ord32_to_f32(r->heap.data[0].key);
        } else {
            if (r->arr_len == r->arr_cap) {
                r->arr_cap = r->arr_cap ? r->arr_cap * 2 : 64;
                r->arr_dist = (float *)realloc(r->arr_dist, r->arr_cap * sizeof(float));
                r->arr_id = (uint32_t *)realloc(r->arr_id, r->arr_cap * sizeof(uint32_t));
            }
            r->arr_dist[r->arr_len] = accurate;
            r->arr_id[r->arr_len] = idx->map_ids[u];
            r->arr_len++;
            r->count++;
            /* f32::max: returns the non-NaN operand */
            if (accurate > r->recent_max_accurate || isnan(r->recent_max_accurate))
                r->recent_max_accurate = accurate;
            if (r->count >= RQO_WINDOW_SIZE) {
                r->threshold = r->recent_max_accurate;
                r->count = 0;
                r->recent_max_accurate = -FLT_MAX;
            }
        }
    }
    g_metrics.precise += precise;
    g_metrics.rough += count;
}

/* ------------------------------------------------------------------------------------------ */
/* src/rabitq.rs: build                                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t key; /* Ord32 of the centroid distance: same order as partial_cmp for non-NaN values */
    uint32_t id;
} label_item_t;

/* stable merge sort on key (rabitq.rs:235 `sort_by` is stable: ties keep ascending original id) */
static void stable_sort_labels(label_item_t *a, label_item_t *tmp, size_t n) {
    if (n < 2) return;
    size_t mid = n / 2;
    stable_sort_labels(a, tmp, mid);
    stable_sort_labels(a + mid, tmp, n - mid);
    size_t i = 0, j = mid, o = 0;
    while (i < mid && j < n) tmp[o++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
    while (i < mid) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, n * sizeof *a);
}

static float *transpose_sq(const float *m, size_t d) {
    float *t = (float *)malloc(d * d * sizeof(float));
    for (size_t r = 0; r < d; ++r)
        for (size_t c = 0; c < d; ++c) t[c * d + r] = m[r * d + c];
    return t;
}

rqo_index_t *rqo_build(const float *base_in, uint64_t n, uint32_t d, const float *centroids_in,
                       uint32_t k, const float *orthogonal) {
    /* rabitq.rs:168-179: zero-pad the dimension to a multiple of 64 */
    uint32_t dim = (d + 63) / 64 * 64;
    rqo_index_t *idx = (rqo_index_t *)calloc(1, sizeof *idx);
    idx->dim = dim;
    idx->n = n;
    idx->k = k;
    float *base = (float *)calloc((size_t)n * dim, sizeof(float));
    float *cent = (float *)calloc((size_t)k * dim, sizeof(float));
    for (uint64_t i = 0; i < n; ++i) memcpy(base + i * dim, base_in + i * d, d * sizeof(float));
    for (uint32_t j = 0; j < k; ++j) memcpy(cent + (size_t)j * dim, centroids_in + (size_t)j * d, d * sizeof(float));

    idx->orthogonal = (float *)malloc((size_t)dim * dim * sizeof(float));
    memcpy(idx->orthogonal, orthogonal, (size_t)dim * dim * sizeof(float));
    idx->orthogonal_t = transpose_sq(orthogonal, dim);

    /* rabitq.rs:188-189: X' = X P, C' = C P (faer GEMM; restated in `project` order) */
    float *xp = (float *)malloc((size_t)n * dim * sizeof(float));
    idx->centroids = (float *)malloc((size_t)k * dim * sizeof(float));
    for (uint64_t i = 0; i < n; ++i) rqo_project(base + i * dim, idx->orthogonal_t, dim, xp + i * dim);
    for (uint32_t j = 0; j < k; ++j)
        rqo_project(cent + (size_t)j * dim, idx->orthogonal_t, dim, idx->centroids + (size_t)j * dim);
    free(cent);

    /* rabitq.rs:192-216 */
    const float dim_sqrt = sqrtf((float)dim);
    uint32_t *label = (uint32_t *)malloc(n * sizeof(uint32_t));
    float *min_dist = (float *)malloc(n * sizeof(float));
    float *x_c_distance = (float *)malloc(n * sizeof(float));
    float *x_dot_product = (float *)malloc(n * sizeof(float));
    float *sign_sum = (float *)malloc(n * sizeof(float));
    rqo_factor_t *factors = (rqo_factor_t *)calloc(n, sizeof(rqo_factor_t));
    const size_t words = dim / 64;
    uint64_t *codes = (uint64_t *)malloc((size_t)n * words * sizeof(uint64_t));
    float *r = (float *)malloc(dim * sizeof(float)), *sgn = (float *)malloc(dim * sizeof(float));
    for (uint64_t i = 0; i < n; ++i) {
        const float *x = xp + i * dim;
        rqo_kmeans_nearest_cluster(idx->centroids, k, dim, x, &label[i], &min_dist[i]);
        const float *c = idx->centroids + (size_t)label[i] * dim;
        for (uint32_t j = 0; j < dim; ++j) r[j] = x[j] - c[j];                      /* :205 */
        float sq = rqo_l2_squared_distance(x, c, dim);
        x_c_distance[i] = sqrtf(sq);                                                /* :206 norm_l2 */
        factors[i].center_distance_square = x_c_distance[i] * x_c_distance[i];      /* :207 powi(2) */
        rqo_vector_binarize_u64(r, dim, codes + i * words);                         /* :208 */
        int pop = 0;
        for (uint32_t j = 0; j < dim; ++j) {                                        /* :209 */
            sgn[j] = r[j] > 0.0f ? 1.0f : -1.0f;
            pop += r[j] > 0.0f;
        }
        sign_sum[i] = (float)(2 * pop - (int)dim);
        float norm = x_c_distance[i] * dim_sqrt;                                    /* :210 */
        if (isnormal(norm))                                                         /* :211-215 */
            x_dot_product[i] = rqo_vector_dot_product(r, sgn, dim) / norm;
        else
            x_dot_product[i] = RQO_DEFAULT_X_DOT_PRODUCT;
    }
    free(r);
    free(sgn);
    free(xp);

    /* rabitq.rs:220-229 */
    const float error_base = 2.0f * RQO_EPSILON / sqrtf((float)dim - 1.0f);
    for (uint64_t i = 0; i < n; ++i) {
        float x_c_over_ip = x_c_distance[i] / x_dot_product[i];
        factors[i].error_bound =
            error_base * sqrtf(x_c_over_ip * x_c_over_ip - factors[i].center_distance_square);
        factors[i].factor_ip = -2.0f / dim_sqrt * x_c_over_ip;
        factors[i].factor_ppc = factors[i].factor_ip * sign_sum[i];
    }

    /* rabitq.rs:232-243: per-cluster stable sort by centroid distance, offsets = prefix sum */
    idx->offsets = (uint32_t *)calloc((size_t)k + 1, sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) idx->offsets[label[i] + 1]++;
    for (uint32_t j = 0; j < k; ++j) idx->offsets[j + 1] += idx->offsets[j];
    label_item_t *items = (label_item_t *)malloc((n ? n : 1) * sizeof *items);
    label_item_t *tmp = (label_item_t *)malloc((n ? n : 1) * sizeof *tmp);
    uint32_t *cursor = (uint32_t *)malloc(((size_t)k + 1) * sizeof(uint32_t));
    memcpy(cursor, idx->offsets, ((size_t)k + 1) * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) { /* ascending id inside each cluster */
        label_item_t it = {rqo_ord32_from_f32(min_dist[i]), (uint32_t)i};
        items[cursor[label[i]]++] = it;
    }
    for (uint32_t j = 0; j < k; ++j)
        stable_sort_labels(items + idx->offsets[j], tmp, idx->offsets[j + 1] - idx->offsets[j]);

    /* rabitq.rs:244-252: gather base / codes / factors into cluster order */
    idx->map_ids = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    idx->base = (float *)malloc((size_t)(n ? n : 1) * dim * sizeof(float));
    idx->x_binary_vec = (uint64_t *)malloc((size_t)(n ? n : 1) * words * sizeof(uint64_t));
    idx->factors = (rqo_factor_t *)malloc((n ? n : 1) * sizeof(rqo_factor_t));
    for (uint64_t p = 0; p < n; ++p) {
        uint32_t id = items[p].id;
        idx->map_ids[p] = id;
        memcpy(idx->base + p * dim, base + (size_t)id * dim, dim * sizeof(float));
        memcpy(idx->x_binary_vec + p * words, codes + (size_t)id * words, words * sizeof(uint64_t));
        idx->factors[p] = factors[id];
    }
    free(items);
    free(tmp);
    free(cursor);
    free(label);
    free(min_dist);
    free(x_c_distance);
    free(x_dot_product);
    free(sign_sum);
    free(factors);
    free(codes);
    free(base);
    return idx;
}

void rqo_free(rqo_index_t *idx) {
    if (!idx) return;
    free(idx->base);
    free(idx->orthogonal);
    free(idx->orthogonal_t);
    free(idx->centroids);
    free(idx->offsets);
    free(idx->map_ids);
    free(idx->x_binary_vec);
    free(idx->factors);
    free(idx);
}

rqo_index_t *rqo_view(uint32_t dim, uint64_t n, uint32_t k, const float *base, const float *orthogonal,
                      const float *centroids, const uint32_t *offsets, const uint32_t *map_ids,
                      const uint64_t *codes, const float *factors) {
    rqo_index_t *idx = (rqo_index_t *)calloc(1, sizeof *idx);
    idx->dim = dim;
    idx->n = n;
    idx->k = k;
    idx->base = (float *)base;
    idx->orthogonal = (float *)orthogonal;
    idx->orthogonal_t = transpose_sq(orthogonal, dim);
    idx->centroids = (float *)centroids;
    idx->offsets = (uint32_t *)offsets;
    idx->map_ids = (uint32_t *)map_ids;
    idx->x_binary_vec = (uint64_t *)codes;
    idx->factors = (rqo_factor_t *)factors;
    return idx;
}

void rqo_free_view(rqo_index_t *idx) {
    if (!idx) return;
    free(idx->orthogonal_t);
    free(idx);
}

/* ------------------------------------------------------------------------------------------ */
/* "vecs" files: [u32 LE count][count x element LE] records (src/utils.rs:280-364)            */
/* ------------------------------------------------------------------------------------------ */
static int write_record(FILE *f, const void *data, uint32_t count, size_t elem) {
    if (fwrite(&count, 4, 1, f) != 1) return -1;
    if (count && fwrite(data, elem, count, f) != count) return -1;
    return 0;
}

static FILE *open_in_dir(const char *dir, const char *name, const char *mode) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    return fopen(path, mode);
}

/* rabitq.rs:128-156 */
int rqo_dump_dir(const rqo_index_t *idx, const char *dir) {
    mkdir(dir, 0777);
    const uint32_t dim = idx->dim;
    FILE *f;
    int rc = 0;
    /* base.fvecs: n records of dim f32, cluster order (:130, base is dim x n, transposed back) */
    if (!(f = open_in_dir(dir, "base.fvecs", "wb"))) return -1;
    for (uint64_t i = 0; i < idx->n && !rc; ++i) rc = write_record(f, idx->base + i * dim, dim, 4);
    fclose(f);
    /* orthogonal.fvecs: dim records, record r = row r of P (:131) */
    if (!(f = open_in_dir(dir, "orthogonal.fvecs", "wb"))) return -1;
    for (uint32_t r = 0; r < dim && !rc; ++r) rc = write_record(f, idx->orthogonal + (size_t)r * dim, dim, 4);
    fclose(f);
    /* centroids.fvecs: the reference's matrix is dim x k, written row-wise => dim records of k (:133) */
    if (!(f = open_in_dir(dir, "centroids.fvecs", "wb"))) return -1;
    float *row = (float *)malloc((size_t)(idx->k ? idx->k : 1) * sizeof(float));
    for (uint32_t r = 0; r < dim && !rc; ++r) {
        for (uint32_t j = 0; j < idx->k; ++j) row[j] = idx->centroids[(size_t)j * dim + r];
        rc = write_record(f, row, idx->k, 4);
    }
    free(row);
    fclose(f);
    /* offsets_ids.ivecs: two records (:136-139) */
    if (!(f = open_in_dir(dir, "offsets_ids.ivecs", "wb"))) return -1;
    if (!rc) rc = write_record(f, idx->offsets, idx->k + 1, 4);
    if (!rc) rc = write_record(f, idx->map_ids, (uint32_t)idx->n, 4);
    fclose(f);
    /* factors.fvecs: ONE record of 4n f32 (:141-149) */
    if (!(f = open_in_dir(dir, "factors.fvecs", "wb"))) return -1;
    if (!rc) rc = write_record(f, idx->factors, (uint32_t)(4 * idx->n), 4);
    fclose(f);
    /* x_binary_vec.u64vecs: ONE record of n*dim/64 u64 (:150-155) */
    if (!(f = open_in_dir(dir, "x_binary_vec.u64vecs", "wb"))) return -1;
    if (!rc) rc = write_record(f, idx->x_binary_vec, (uint32_t)(idx->n * (dim / 64)), 8);
    fclose(f);
    return rc;
}

/* reads every record of a vecs file into one flat buffer; returns record count and the length of
 * the first/last record. */
static void *read_all_records(const char *dir, const char *name, size_t elem, uint64_t *nrec,
                              uint32_t *first_len, uint32_t *last_len, uint64_t *total) {
    FILE *f = open_in_dir(dir, name, "rb");
    if (!f) return NULL;
    size_t cap = 1 << 16, used = 0;
    char *buf = (char *)malloc(cap);
    uint32_t cnt;
    *nrec = 0;
    *total = 0;
    *first_len = *last_len = 0;
    while (fread(&cnt, 4, 1, f) == 1) {
        size_t bytes = (size_t)cnt * elem;
        while (used + bytes > cap) buf = (char *)realloc(buf, cap *= 2);
        if (bytes && fread(buf + used, 1, bytes, f) != bytes) {
            free(buf);
            fclose(f);
            return NULL;
        }
        if (*nrec == 0) *first_len = cnt;
        *last_len = cnt;
        used += bytes;
        *total += cnt;
        ++*nrec;
    }
    fclose(f);
    return buf;
}

/* rabitq.rs:84-125 */
rqo_index_t *rqo_load_dir(const char *dir) {
    uint64_t nrec, total;
    uint32_t fl, ll;
    rqo_index_t *idx = (rqo_index_t *)calloc(1, sizeof *idx);
    idx->orthogonal = (float *)read_all_records(dir, "orthogonal.fvecs", 4, &nrec, &fl, &ll, &total);
    if (!idx->orthogonal) goto fail;
    idx->dim = (uint32_t)nrec; /* :108 dim = orthogonal.nrows() */
    if (idx->dim % 64 != 0) goto fail;
    idx->orthogonal_t = transpose_sq(idx->orthogonal, idx->dim);
    {
        float *ct = (float *)read_all_records(dir, "centroids.fvecs", 4, &nrec, &fl, &ll, &total);
        if (!ct) goto fail;
        idx->k = fl; /* dim records of k values */
        idx->centroids = (float *)malloc((size_t)(idx->k ? idx->k : 1) * idx->dim * sizeof(float));
        for (uint32_t r = 0; r < idx->dim; ++r)
            for (uint32_t j = 0; j < idx->k; ++j)
                idx->centroids[(size_t)j * idx->dim + r] = ct[(size_t)r * idx->k + j];
        free(ct);
    }
    {
        uint32_t *oi = (uint32_t *)read_all_records(dir, "offsets_ids.ivecs", 4, &nrec, &fl, &ll, &total);
        if (!oi) goto fail;
        idx->offsets = (uint32_t *)malloc((size_t)fl * 4);
        memcpy(idx->offsets, oi, (size_t)fl * 4);                 /* .first() */
        idx->map_ids = (uint32_t *)malloc((size_t)(ll ? ll : 1) * 4);
        memcpy(idx->map_ids, oi + (total - ll), (size_t)ll * 4);  /* .last()  */
        idx->n = ll;
        free(oi);
    }
    idx->factors = (rqo_factor_t *)read_all_records(dir, "factors.fvecs", 4, &nrec, &fl, &ll, &total);
    if (!idx->factors) goto fail;
    idx->x_binary_vec = (uint64_t *)read_all_records(dir, "x_binary_vec.u64vecs", 8, &nrec, &fl, &ll, &total);
    if (!idx->x_binary_vec) goto fail;
    idx->base = (float *)read_all_records(dir, "base.fvecs", 4, &nrec, &fl, &ll, &total);
    if (!idx->base) goto fail;
    return idx;
fail:
    rqo_free(idx);
    return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* src/rabitq.rs: query                                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t key;
    uint32_t cluster;
} coarse_item_t;

static int coarse_cmp(const void *a, const void *b) {
    const coarse_item_t *x = (const coarse_item_t *)a, *y = (const coarse_item_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->cluster < y->cluster ? -1 : (x->cluster > y->cluster);
}

/* rabitq.rs:283-297.  total_cmp order == Ord32 order.  select_nth_unstable + stable sort leave the
 * order of EXACTLY equal distances unspecified in the reference; this oracle (and the GPU engine)
 * break such ties by ascending cluster id. */
int rqo_coarse_rank(const rqo_index_t *idx, const float *y, uint32_t probe, uint32_t *out_cluster,
                    float *out_dist) {
    uint32_t k = idx->k;
    uint32_t length = probe < k ? probe : k;
    if (length == 0) return -1; /* rabitq.rs:295 underflows `length - 1` and panics */
    coarse_item_t *lists = (coarse_item_t *)malloc((size_t)k * sizeof *lists);
    for (uint32_t i = 0; i < k; ++i) {
        float d = rqo_l2_squared_distance(idx->centroids + (size_t)i * idx->dim, y, idx->dim);
        lists[i].key = rqo_ord32_from_f32(d);
        lists[i].cluster = i;
    }
    qsort(lists, k, sizeof *lists, coarse_cmp);
    for (uint32_t i = 0; i < length; ++i) {
        out_cluster[i] = lists[i].cluster;
        out_dist[i] = rqo_ord32_to_f32(lists[i].key);
    }
    free(lists);
    return (int)length;
}

/* rabitq.rs:304-317 */
void rqo_query_prep(const rqo_index_t *idx, const float *y, uint32_t cluster, float *out_lower,
                    float *out_delta, uint32_t *out_sum, uint64_t *out_planes) {
    const uint32_t dim = idx->dim;
    float *residual = (float *)malloc(dim * sizeof(float));
    uint8_t *quantized = (uint8_t *)calloc(dim, 1);
    float lo, hi;
    rqo_min_max_residual(residual, y, idx->centroids + (size_t)cluster * dim, dim, &lo, &hi);
    const float scalar = 1.0f / 15.0f; /* consts.rs:10 */
    float delta = (hi - lo) * scalar;
    float one_over_delta = 1.0f / delta; /* f32::recip */
    *out_sum = rqo_scalar_quantize(quantized, residual, dim, lo, one_over_delta);
    memset(out_planes, 0, (size_t)(dim / 64) * RQO_THETA_LOG_DIM * sizeof(uint64_t));
    rqo_vector_binarize_query(quantized, dim, out_planes);
    *out_lower = lo;
    *out_delta = delta;
    free(residual);
    free(quantized);
}

/* rabitq.rs:336-367.  Expression order (left to right, no contraction):
 *   ((cds + ycd) + lo*ppc) + (((2*s - sumq) * fip) * delta)  -  eb * sqrt(ycd) */
void rqo_scan_cluster(const rqo_index_t *idx, uint32_t cluster, float y_c_distance_square,
                      const uint64_t *planes, float lower_bound, float scalar_sum, float delta,
                      float *out_rough) {
    const size_t words = idx->dim / 64;
    const float dist_sqrt = sqrtf(y_c_distance_square);
    const uint32_t lo = idx->offsets[cluster], hi = idx->offsets[cluster + 1];
    for (uint32_t j = lo; j < hi; ++j) {
        const rqo_factor_t *f = &idx->factors[j];
        float s = (float)rqo_asymmetric_binary_dot_product(idx->x_binary_vec + (size_t)j * words, planes, words);
        float t = f->center_distance_square + y_c_distance_square;
        t = t + lower_bound * f->factor_ppc;
        t = t + (2.0f * s - scalar_sum) * f->factor_ip * delta;
        t = t - f->error_bound * dist_sqrt;
        out_rough[j - lo] = t;
    }
}

static uint32_t max_list_len(const rqo_index_t *idx) {
    uint32_t m = 0;
    for (uint32_t c = 0; c < idx->k; ++c) {
        uint32_t l = idx->offsets[c + 1] - idx->offsets[c];
        if (l > m) m = l;
    }
    return m;
}

static int heur_cmp(const void *a, const void *b) {
    const label_item_t *x = (const label_item_t *)a, *y = (const label_item_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

int rqo_query(const rqo_index_t *idx, const float *query, uint32_t len, uint32_t probe, uint32_t topk,
              int heuristic_rank, float *out_dist, uint32_t *out_id, uint32_t *out_n) {
    const uint32_t dim = idx->dim;
    *out_n = 0;
    if (dim != (len + 63) / 64 * 64) return -2; /* rabitq.rs:275 assert_eq! */
    float *q = (float *)calloc(dim, sizeof(float));
    memcpy(q, query, len * sizeof(float)); /* :277-280 zero pad */
    float *y = (float *)malloc(dim * sizeof(float));
    rqo_project(q, idx->orthogonal_t, dim, y); /* :282 */
    uint32_t cap = probe < idx->k ? probe : idx->k;
    uint32_t *clusters = (uint32_t *)malloc((cap ? cap : 1) * sizeof(uint32_t));
    float *cdist = (float *)malloc((cap ? cap : 1) * sizeof(float));
    int length = rqo_coarse_rank(idx, y, probe, clusters, cdist);
    int rc = 0;
    if (length < 0) {
        rc = -3;
        goto done;
    }
    {
        reranker_t rr;
        reranker_init(&rr, q, topk, heuristic_rank); /* :299 un-rotated, padded query */
        uint64_t *planes = (uint64_t *)malloc((size_t)(dim / 64) * RQO_THETA_LOG_DIM * sizeof(uint64_t));
        float *rough = (float *)malloc((size_t)(max_list_len(idx) + 1) * sizeof(float));
        for (int s = 0; s < length; ++s) { /* :304-329 */
            float lo, delta;
            uint32_t sumq;
            uint32_t c = clusters[s];
            rqo_query_prep(idx, y, c, &lo, &delta, &sumq, planes);
            rqo_scan_cluster(idx, c, cdist[s], planes, lo, (float)sumq, delta, rough);
            rank_batch(&rr, rough, idx->offsets[c], idx->offsets[c + 1] - idx->offsets[c], idx);
        }
        g_metrics.query += 1; /* :331 */
        if (!heuristic_rank) { /* rerank.rs:108-113: heap Vec order */
            for (size_t i = 0; i < rr.heap.len; ++i) {
                out_dist[i] = rqo_ord32_to_f32(rr.heap.data[i].key);
                out_id[i] = rr.heap.data[i].id;
            }
            *out_n = (uint32_t)rr.heap.len;
        } else { /* rerank.rs:170-176: the topk smallest by total_cmp; order unspecified -> sorted */
            size_t length2 = topk < rr.arr_len ? topk : rr.arr_len;
            if (length2 == 0) {
                rc = -4; /* `length - 1` underflow panic */
            } else {
                label_item_t *items = (label_item_t *)malloc(rr.arr_len * sizeof *items);
                for (size_t i = 0; i < rr.arr_len; ++i) {
                    items[i].key = rqo_ord32_from_f32(rr.arr_dist[i]);
                    items[i].id = (uint32_t)i;
                }
                qsort(items, rr.arr_len, sizeof *items, heur_cmp);
                for (size_t i = 0; i < length2; ++i) {
                    out_dist[i] = rr.arr_dist[items[i].id];
                    out_id[i] = rr.arr_id[items[i].id];
                }
                *out_n = (uint32_t)length2;
                free(items);
            }
        }
        free(planes);
        free(rough);
        reranker_free(&rr);
    }
done:
    free(clusters);
    free(cdist);
    free(q);
    free(y);
    return rc;
}

uint64_t rqo_scan_only(const rqo_index_t *idx, const float *query, uint32_t len, uint32_t probe,
                       float *scratch_rough) {
    const uint32_t dim = idx->dim;
    float *q = (float *)calloc(dim, sizeof(float));
    memcpy(q, query, len * sizeof(float));
    float *y = (float *)malloc(dim * sizeof(float));
    rqo_project(q, idx->orthogonal_t, dim, y);
    uint32_t cap = probe < idx->k ? probe : idx->k;
    uint32_t *clusters = (uint32_t *)malloc((cap ? cap : 1) * sizeof(uint32_t));
    float *cdist = (float *)malloc((cap ? cap : 1) * sizeof(float));
    int length = rqo_coarse_rank(idx, y, probe, clusters, cdist);
    uint64_t *planes = (uint64_t *)malloc((size_t)(dim / 64) * RQO_THETA_LOG_DIM * sizeof(uint64_t));
    uint64_t scanned = 0;
    for (int s = 0; s < length; ++s) {
        float lo, delta;
        uint32_t sumq, c = clusters[s];
        rqo_query_prep(idx, y, c, &lo, &delta, &sumq, planes);
        rqo_scan_cluster(idx, c, cdist[s], planes, lo, (float)sumq, delta, scratch_rough);
        scanned += idx->offsets[c + 1] - idx->offsets[c];
    }
    free(planes);
    free(clusters);
    free(cdist);
    free(q);
    free(y);
    return scanned;
}
