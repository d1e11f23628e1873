/*
 * rabitq_hip.h -- C ABI of librabitq_hip.so, the MI355X (gfx950) engine for the RaBitQ build/query
 * hot path of kemingy/rabitq.  Plain pointers and sizes only; no C++/torch types.
 *
 * The reference has no FFI seam of its own; the seam is cut at the public
 * methods of `RaBitQ` (src/lib.rs:12), which is exactly what a Rust `extern "C"` block would bind
 * (INTEGRATION.md shows that binding).  Every entry point cites the reference interface it
 * replaces as file:line relative to the reference repository root.
 *
 * Conventions
 *   - All functions return rq_status (0 = RQ_OK).  The reference panics (`expect`/`assert!`,
 *     release profile panic = "abort", Cargo.toml:43) where these return an error; a Rust wrapper
 *     keeps that behaviour by `expect()`-ing the status.  rq_last_error() gives the message.
 *   - "host" pointers are ordinary CPU memory; "_device" variants take HIP device pointers
 *     (hipMalloc) and enqueue on an internal stream, returning after the results are complete.
 *   - Query entry points are safe to call concurrently on one index (read-only index, per-call
 *     workspace + stream), as `RaBitQ::query(&self)` is (src/rabitq.rs:268); build/load/dump/free
 *     are single-caller.
 *   - Matrices are row-major.  `dim` is the dimension padded to a multiple of 64
 *     (src/rabitq.rs:168-179).  Ids and positions are u32 (src/rabitq.rs:64-65).
 */
#ifndef RABITQ_HIP_H
#define RABITQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rq_status;
enum {
    RQ_OK = 0,
    RQ_ERR_INVALID = -1,      /* bad argument (null pointer, probe == 0, topk == 0 ...)            */
    RQ_ERR_DIM_MISMATCH = -2, /* src/rabitq.rs:165 / :275 assert failures                          */
    RQ_ERR_IO = -3,           /* src/rabitq.rs:73-155 `expect("open ... error")`                    */
    RQ_ERR_HIP = -4,          /* a HIP runtime call failed                                         */
    RQ_ERR_NO_DEVICE = -5,    /* no gfx950 device visible: there is NO CPU fallback                */
    RQ_ERR_UNSUPPORTED = -6,  /* size outside the engine's limits (documented per function)        */
    RQ_ERR_EMPTY = -7,        /* heuristic ranker produced no candidate (src/rerank.rs:173 panics) */
    RQ_ERR_OOM = -8
};

/* Opaque index handle: the device-resident state of `RaBitQ` (src/rabitq.rs:57-68). */
typedef struct rq_index rq_index;

/* `Factor`, src/rabitq.rs:21-32 (repr(C), 16 bytes, same field order). */
typedef struct {
    float factor_ip, factor_ppc, error_bound, center_distance_square;
} rq_factor_t;

/* `Metrics`, src/metrics.rs:7-18: process-global counters. */
typedef struct {
    uint64_t rough, precise, query, miss;
} rq_metrics_t;

/* Out-structs (rq_info_t, rq_build_stats_t, rq_profile_t) are versioned by size: the caller sets `struct_size` to
 * sizeof(the struct it was compiled against) before the call and the library writes at most that many bytes, so a
 * host built against an older header is never written past its struct when fields are appended (fields are only ever
 * appended).  struct_size == 0 (an unset field) is refused with RQ_ERR_INVALID. */
typedef struct {
    uint32_t struct_size;  /* in: sizeof(rq_info_t) of the caller */
    uint32_t dim;          /* padded dimension */
    uint32_t k;            /* number of clusters */
    uint32_t max_list_len; /* longest IVF list */
    uint64_t n;            /* number of vectors */
    uint64_t n_hbm;        /* raw vectors resident in HBM; the other n - n_hbm (list tails) are in pinned host memory */
    uint32_t split_rows;   /* 1: the raw vectors are stored as split rows (option "split_rows"); appended in 0.5.0 */
    uint32_t reserved0;
} rq_info_t;

/* ---- library ------------------------------------------------------------------------------- */
/* ABI revision of this header: bumped whenever a struct layout or the meaning of an entry point changes (3: sized
 * out-structs, rq_get_device_ptr refuses RQ_ARR_BASE on tiered indexes; options "scan_dense" and the round-2 "coarse_impl" = 3 removed.
 * 4: rq_set_option("scan_debug") refuses the timing-ablation bits -- they exist in the developer build only --, the matrix-core
 * scan's step counters in rq_profile_t are always filled, new option "scan_gate", "coarse_impl" = 3 is back with a new meaning,
 * rq_profile_t.reserved became coarse_fallback_rows, .reserved2 matrix_additive_launches; later in revision 4, additions only: "coarse_impl" = 4,
 * "coarse_tiled_from", "rerank_shadow" = 2 -- the new default; 0.5.0: option "split_rows", rq_info_t.split_rows appended, RQ_ARR_BASE
 * refused for split rows as for tiers).  A host checks rq_abi_version() ==
 * RQ_ABI_VERSION once after loading the library. */
#define RQ_ABI_VERSION 4
uint32_t rq_abi_version(void);
const char *rq_version(void);
const char *rq_last_error(void);              /* thread-local message of the last failure        */
rq_status rq_init(int device);                /* select the HIP device for this process          */

/* ---- build: RaBitQ::from_path, src/rabitq.rs:159-265 ---------------------------------------- */
/* `orthogonal` is the dim x dim rotation P (row-major, P[r][c]); the reference draws it from an
 * unseeded RNG (src/utils.rs:16-20), so for reproducibility it is an input.  NULL = generate a
 * Gaussian-QR orthogonal matrix from `seed` (same construction, seeded). */
rq_status rq_build(const float *base, uint64_t n, uint32_t d, const float *centroids, uint32_t k,
                   const float *orthogonal, uint64_t seed, rq_index **out);
/* Same with base/centroids already in device memory (n x d and k x d, row-major f32).  `base` is
 * only read; the index keeps its own cluster-ordered copy (src/rabitq.rs:245-247). */
rq_status rq_build_device(const float *d_base, uint64_t n, uint32_t d, const float *d_centroids,
                          uint32_t k, const float *orthogonal_host, uint64_t seed, rq_index **out);
/* From .fvecs files exactly as RaBitQ::from_path(base_path, centroid_path). */
rq_status rq_build_from_path(const char *base_fvecs, const char *centroid_fvecs,
                             const float *orthogonal, uint64_t seed, rq_index **out);

/* ---- streamed two-pass build: from_path for inputs that need not be resident (e.g. 100M x 768 = 307 GB) ---- */
/* The same result as rq_build_device, with the n x d input fed chunk by chunk, twice:
 *   rq_builder_create -> rq_builder_assign_chunk (every row once: rotate src/rabitq.rs:188, nearest list :203, sign-pack +
 *   factors :205-229) -> rq_builder_order (cluster ordering :232-243) -> rq_builder_place_chunk (every row once more:
 *   raw vectors to their cluster-order positions :244-247) -> rq_builder_finish.
 * Chunks are m x d row-major f32 in DEVICE memory covering rows [i0, i0 + m); any order, any sizes, but every row exactly
 * once per pass: a chunk that overlaps rows already fed to the same pass is refused (RQ_ERR_INVALID), and
 * rq_builder_order / rq_builder_finish refuse to run while rows are missing.  Each call returns
 * when the chunk buffer may be reused.  max_device_base_bytes: HBM budget of the raw vectors (0 = automatic: what is
 * free minus a reserve; UINT64_MAX = all in HBM).  Beyond it the split is per list: the head of every list (the vectors
 * nearest its centroid, which the re-ranker asks for most) stays in HBM, the tail goes to pinned host memory and is
 * gathered over the host link by the rerank (results identical).  rq_builder_finish consumes the builder (also on
 * error); rq_builder_free abandons one. */
typedef struct rq_builder rq_builder;
typedef struct {
    uint32_t struct_size;                      /* in: sizeof(rq_build_stats_t) of the caller */
    float ms_rotate, ms_assign, ms_quantize;   /* device time of the three pass-1 kernels, HIP events, summed over chunks */
    uint64_t rows_assigned;                    /* rotation flops so far = 2 * rows_assigned * dim^2 */
    uint64_t rows_in_hbm, rows_in_host_memory; /* base tiers (known after rq_builder_order) */
    uint64_t rows_exact_redo;                  /* rows whose nearest list the exact-order kernel decided alone: the bf16 pre-filter
                                                  left none or more than 16 candidates (appended in revision 3) */
} rq_build_stats_t;
rq_status rq_builder_create(uint64_t n, uint32_t d, const float *d_centroids, uint32_t k, const float *orthogonal_host,
                            uint64_t seed, uint64_t max_device_base_bytes, rq_builder **out);
rq_status rq_builder_assign_chunk(rq_builder *b, const float *d_rows, uint64_t i0, uint64_t m);
rq_status rq_builder_order(rq_builder *b);
rq_status rq_builder_place_chunk(rq_builder *b, const float *d_rows, uint64_t i0, uint64_t m);
rq_status rq_builder_finish(rq_builder *b, rq_index **out);
void rq_builder_free(rq_builder *b);
rq_status rq_builder_stats(const rq_builder *b, rq_build_stats_t *out);

/* ---- centroid training (scripts/cluster.py:63-108 does this offline with faiss k-means) -------- */
/* Lloyd k-means on a sample of min(n, points_per_centroid * k) vectors (points_per_centroid = 256
 * in the reference script); d_base is n x d in device memory, d_centroids_out receives k x d.
 * No parity target (faiss is not available; results depend on the seed): judged by recall. */
rq_status rq_kmeans_device(const float *d_base, uint64_t n, uint32_t d, uint32_t k, uint32_t iters,
                           uint32_t points_per_centroid, uint64_t seed, float *d_centroids_out);

/* ---- persistence: load_from_dir / dump_to_dir, src/rabitq.rs:84-156 -------------------------- */
/* Byte-compatible with the crate's five-file directory (vecs framing: src/utils.rs:280-364). */
rq_status rq_load_dir(const char *dir, rq_index **out);
rq_status rq_dump_dir(const rq_index *idx, const char *dir);
/* load_from_json / dump_to_json, src/rabitq.rs:72-81: the serde_json image of the `RaBitQ` struct (faer `Mat`s as
 * {"nrows","ncols","data": row-major}; base dim x n, centroids dim x k).  A debugging format in the reference too: the
 * text is ~10x the binary directory, so use rq_load_dir / rq_dump_dir for anything large. */
rq_status rq_load_json(const char *path, rq_index **out);
rq_status rq_dump_json(const rq_index *idx, const char *path);
void rq_free(rq_index *idx);

/* Construct an index from the reference's in-memory arrays (what load_from_dir ends up with):
 * base n x dim (cluster order, un-rotated), orthogonal dim x dim, centroids k x dim (rotated;
 * row j = centroid j), offsets k+1, map_ids n, codes n x dim/64 u64, factors n x 4 f32. */
rq_status rq_from_arrays(uint32_t dim, uint64_t n, uint32_t k, const float *base,
                         const float *orthogonal, const float *centroids, const uint32_t *offsets,
                         const uint32_t *map_ids, const uint64_t *codes, const rq_factor_t *factors,
                         rq_index **out);

rq_status rq_info(const rq_index *idx, rq_info_t *out);
/* Copy one array of the index back to host memory (sizes as in rq_from_arrays). */
enum { RQ_ARR_BASE = 0, RQ_ARR_ORTHOGONAL, RQ_ARR_CENTROIDS, RQ_ARR_OFFSETS, RQ_ARR_MAP_IDS,
       RQ_ARR_CODES, RQ_ARR_FACTORS };
rq_status rq_get_array(const rq_index *idx, int which, void *dst, uint64_t dst_bytes);
/* Device pointer of one array (valid until rq_free); for zero-copy hand-over to a caller that
 * already lives on the GPU.  RQ_ARR_BASE: every row at its position -- only when n_hbm == n; on a tiered index (list
 * tails in pinned host memory) or one that stores its raw vectors as split rows (option "split_rows") there is no single f32
 * device array and the call returns RQ_ERR_UNSUPPORTED (use rq_get_array for a host copy of the whole array). */
rq_status rq_get_device_ptr(const rq_index *idx, int which, const void **out_ptr, uint64_t *out_bytes);

/* ---- query: RaBitQ::query, src/rabitq.rs:268-333 --------------------------------------------- */
/* One query of `len` floats (dim must equal ceil(len/64)*64, :275).  out_dist/out_id need room for
 * topk entries; *out_n receives the number written (<= topk).  Heap ranker results come in the
 * reference's heap-internal order (src/rerank.rs:108-113); heuristic results sorted ascending. */
rq_status rq_query(const rq_index *idx, const float *query, uint32_t len, uint32_t probe,
                   uint32_t topk, int heuristic_rank, float *out_dist, uint32_t *out_id,
                   uint32_t *out_n);
/* B independent queries (row-major B x len); the form the GPU wants.  Results of query b are at
 * out_dist/out_id[b*topk ...], count in out_n[b].  Each query's result is identical to rq_query's. */
rq_status rq_query_batch(const rq_index *idx, const float *queries, uint32_t nq, uint32_t len,
                         uint32_t probe, uint32_t topk, int heuristic_rank, float *out_dist,
                         uint32_t *out_id, uint32_t *out_n);
/* Same with queries and outputs in device memory.  Every entry point that takes device pointers runs on
 * HIP streams of its own (non-blocking with respect to the caller's streams) and returns with its outputs
 * complete: inputs produced on another stream must be complete before the call (synchronise that stream). */
rq_status rq_query_batch_device(const rq_index *idx, const float *d_queries, uint32_t nq,
                                uint32_t len, uint32_t probe, uint32_t topk, int heuristic_rank,
                                float *d_out_dist, uint32_t *d_out_id, uint32_t *d_out_n);
/* The same call in two halves, for callers that keep several batches in flight (a serving loop):
 * _begin enqueues the whole batch on a workspace / HIP stream of its own and returns a ticket; _end
 * waits for it, performs the (rare) survivor-buffer re-runs and the counter updates, reports the
 * call's status and frees the ticket.  Queries and outputs must stay valid and untouched in between.
 * Batches begun back to back overlap on the device: one batch's HBM-bound stages (exact rerank) run
 * beside another's compute-bound scan.  Every begun ticket must be ended exactly once.
 * Exceptions to "returns at once": (i) a batch that needs several passes (more queries than one pass holds) runs all but its last
 * pass inside _begin; (ii) on a shard-like index -- fewer than half of its lists have members, i.e. a shard carved by
 * rq_shard_index -- a pass of >= 65 536 (query, list) pairs sizes its launches by the number of pairs whose list lives here, which
 * it reads back once: _begin then returns after rotate, coarse ranking and that split have run (tens of microseconds of device
 * work), so two such batches overlap from the query quantisation on. */
typedef struct rq_ticket rq_ticket;
rq_status rq_query_batch_device_begin(const rq_index *idx, const float *d_queries, uint32_t nq,
                                      uint32_t len, uint32_t probe, uint32_t topk, int heuristic_rank,
                                      float *d_out_dist, uint32_t *d_out_id, uint32_t *d_out_n,
                                      rq_ticket **out_ticket);
rq_status rq_query_batch_device_end(rq_ticket *ticket);

/* ---- sharded deployments (one index shard per GPU / process) ----------------------------------- */
/* The coarse ranking of src/rabitq.rs:283-297 restricted to lists [list_lo, list_hi): the `probe`
 * nearest of them per query, ascending, as (global list id, distance); rows are `probe` wide, padded
 * with (0xFFFFFFFF, +inf) when the range holds fewer lists.  A shard ranks only the lists it owns;
 * the per-shard rows are all-gathered and merged by the caller (rabitq_amd/sharding.py). */
rq_status rq_coarse_topk_device(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                uint32_t list_lo, uint32_t list_hi, uint32_t probe, uint32_t *d_out_cluster,
                                float *d_out_dist);
/* Shard merge: per query the m_out smallest of the world x width u64 keys in d_in[world][nq][width] (rows in
 * any order), ascending, into d_out[nq][m_out] (padded with ~0).  Keys: (f32 distance bits << 32 | list id) for
 * probe lists, (Ord32 image << 32 | global id) for per-shard top-k.  Launched asynchronously on the legacy
 * default stream (ordered with the caller's default-stream work, e.g. the all-gather that produced d_in). */
rq_status rq_merge_smallest_u64_device(const uint64_t *d_in, uint32_t world, uint32_t nq, uint32_t width,
                                       uint32_t m_out, uint64_t *d_out);
/* rq_query_batch_device with the probe lists supplied by the caller (nq x probe, visiting order,
 * 0xFFFFFFFF = no list; probe <= k) instead of ranked internally. */
rq_status rq_query_batch_device_probed(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                       const uint32_t *d_probe_cluster, const float *d_probe_dist, uint32_t probe,
                                       uint32_t topk, int heuristic_rank, float *d_out_dist, uint32_t *d_out_id,
                                       uint32_t *d_out_n);

/* The same with a per-query INITIAL threshold (device, nq floats; f32::MAX = none): a candidate is re-ranked only if
 * rough < threshold and kept only if accurate < threshold, from the first candidate on -- the multi-GPU step seeds every
 * shard with the k-th best distance the owner of the query's nearest list found there (one all-reduce(min)), so that
 * a shard which does not hold a query's neighbourhood stops re-ranking its far candidates (per-shard
 * thresholds).  The whole candidate stream runs as one stage.  A query may return fewer than topk results: everything
 * left out is at or above its initial threshold.  No reference counterpart (single process). */
rq_status rq_query_batch_device_seeded(const rq_index *idx, const float *d_queries, uint32_t nq, uint32_t len,
                                       const uint32_t *d_probe_cluster, const float *d_probe_dist, uint32_t probe,
                                       uint32_t topk, int heuristic_rank, const float *d_thr_init, float *d_out_dist,
                                       uint32_t *d_out_id, uint32_t *d_out_n);

/* List partitioner: whole IVF lists to `world` shards, greedy by list length (longest first,
 * each to the least-loaded shard; deterministic).  out_owner[c] = shard of list c (k entries, host);
 * out_load (world entries, host, may be NULL) = vectors per shard.  The reference has no counterpart (single process). */
rq_status rq_partition_lists(const rq_index *idx, uint32_t world, uint32_t *out_owner, uint64_t *out_load);
/* Carve shard `rank` out of an index: a new index with the same dim / k / rotation / (replicated) centroids that
 * holds only the lists with owner[c] == rank (the others are empty); map_ids keep the ORIGINAL ids, so per-shard
 * results are global.  The source index is unchanged; free both with rq_free. */
rq_status rq_shard_index(const rq_index *idx, const uint32_t *owner, uint32_t rank, rq_index **out);
/* The whole multi-GPU step behind the C ABI, for hosts without torch.  Every rank holds all (replicated) rotated centroids
 * and a subset of the lists.  `nccl_comm` is an ncclComm_t of `world` ranks created by the caller with ncclCommInitRank
 * (may be NULL when world == 1).  The step, all on one HIP stream of the engine:
 *   handshake  one ncclAllReduce(max) of three int32 (a hash of the call's parameters and an error flag);
 *   coarse     rank r ranks lists [r k / world, (r+1) k / world) only, one ncclAllGather of the nq x probe
 *              (distance, list) keys + merge: the same global probe list on every rank;
 *   answer     the shard's own lists of that probe list, with thresholds shared between the shards (option
 *              "shared_thresholds": the nearest list first, one ncclAllReduce(min) of the k-th best distances found
 *              there, the rest seeded with it; see rq_query_batch_device_seeded) or on the shard's own thresholds;
 *   merge      ONE ncclAllGather of the per-shard top-k as u64 keys (+ a status word per rank), k-way merge.
 * Every rank receives the same global top-k, ascending by (distance, id); ids are shard-local map_ids + id_offset (u32).
 * COLLECTIVE CONTRACT: all ranks of the communicator call this entry with the same nq, len, probe, topk, world,
 * heuristic_rank and the same "shared_thresholds" option, in the same order.  The handshake detects a violation (and a
 * rank that failed validation or allocation) and makes every rank return an error before any data collective; after
 * it a rank that fails locally keeps taking part in every collective and reports through its status word, so no peer is
 * left blocked and all ranks return an error for that step.  Limits: world * topk <= 8192, nq * topk < 2^32.
 * RCCL is resolved at first use from the host process (or RABITQ_RCCL_LIB), so the library itself does not link against it.
 * rq_last_profile() afterwards holds the sum over the step's engine passes. */
rq_status rq_query_batch_sharded_device(const rq_index *shard, void *nccl_comm, uint32_t world, uint32_t id_offset,
                                        const float *d_queries, uint32_t nq, uint32_t len, uint32_t probe, uint32_t topk,
                                        int heuristic_rank, float *d_out_dist, uint32_t *d_out_id, uint32_t *d_out_n);
/* The collectives rq_query_batch_sharded_device calls, for hosts with another transport than RCCL (or a test harness):
 * same signatures and semantics as ncclAllGather / ncclAllReduce / ncclCommUserRank (device buffers, enqueued on
 * `hip_stream` or completed before returning; dtype / op are the ncclDataType_t / ncclRedOp_t values: int32 = 2,
 * uint64 = 5, float32 = 7; max = 2, min = 3; return 0 on success).  `comm` is passed through untouched.  NULL = back to
 * RCCL (the default).  Process-wide; set it before the first sharded call. */
typedef struct {
    uint32_t struct_size; /* in: sizeof(rq_collectives_t) */
    uint32_t reserved;
    int (*all_gather)(const void *send, void *recv, size_t send_count, int dtype, void *comm, void *hip_stream);
    int (*all_reduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, void *hip_stream);
    int (*comm_user_rank)(void *comm, int *rank);
} rq_collectives_t;
rq_status rq_set_collectives(const rq_collectives_t *collectives);

/* ---- metrics: METRICS, src/metrics.rs:65 ------------------------------------------------------ */
rq_status rq_metrics(rq_metrics_t *out);
rq_status rq_metrics_reset(void);

/* ---- per-stage entry points (each hot kernel testable in isolation; all host pointers) -------- */
/* Rotation X' = X P in the reference's `project` order (src/utils.rs:237-258, src/simd.rs:257-314):
 * x is n x dim, out is n x dim.  use_mfma = 1 runs the MFMA kernel (bit-identical by construction),
 * 0 the VALU kernel. */
rq_status rq_rotate(const float *x, uint64_t n, uint32_t dim, const float *orthogonal, int use_mfma,
                    float *out);
/* Same rotation on device-resident rows with the index's own P (the build's rotate step,
 * src/rabitq.rs:188): d_x and d_out are n x dim device arrays.  If out_ms is non-NULL it receives
 * the MFMA kernel's duration measured with HIP events on the launch stream (bench.py's rotation
 * roofline). */
rq_status rq_rotate_device(const rq_index *idx, const float *d_x, uint64_t n, float *d_out, float *out_ms);
/* Nearest rotated centroid + residual sign-pack + factors for already-rotated vectors
 * (src/utils.rs:261-277, :53-67; src/rabitq.rs:203-229): out_label n, out_dist n,
 * out_codes n x dim/64, out_factors n. */
rq_status rq_quantize_pack(const float *x_rot, uint64_t n, uint32_t dim, const float *centroids_rot,
                           uint32_t k, uint32_t *out_label, float *out_dist, uint64_t *out_codes,
                           rq_factor_t *out_factors);
/* Rotate + coarse-rank a query batch (src/rabitq.rs:282-297): out_y nq x dim (may be NULL),
 * out_cluster / out_dist nq x min(probe,k). */
rq_status rq_coarse_rank(const rq_index *idx, const float *queries, uint32_t nq, uint32_t len,
                         uint32_t probe, float *out_y, uint32_t *out_cluster, float *out_dist);
/* Per-(query,cluster) query quantisation (src/rabitq.rs:304-317; src/simd.rs:117-247, :83-107):
 * y is nq x dim rotated queries, cluster[i] is the cluster paired with y row i.
 * out_planes nq x 4*dim/64 u64. */
rq_status rq_query_prep(const rq_index *idx, const float *y, uint32_t nq, const uint32_t *cluster,
                        float *out_lower, float *out_delta, uint32_t *out_sum, uint64_t *out_planes);
/* calculate_rough_distance (src/rabitq.rs:336-367) for the whole list of `cluster` with the
 * given query-side scalars and bit planes; out_rough needs list-length floats. */
rq_status rq_scan(const rq_index *idx, uint32_t cluster, float y_c_distance_square,
                  const uint64_t *planes, float lower_bound, float scalar_sum, float delta,
                  float *out_rough);
/* Exact f32 L2 rerank distance (src/simd.rs:14-73 order) of the padded, un-rotated query against
 * base rows at cluster-order positions pos[0..m) (src/rerank.rs:85-90). */
rq_status rq_rerank(const rq_index *idx, const float *query_padded, const uint32_t *pos, uint32_t m,
                    float *out_accurate);

/* ---- instrumentation ------------------------------------------------------------------------- */
/* Per-kernel device time of the LAST rq_query_batch* call on this thread, measured with HIP events
 * on the stream the kernels ran on.  Names: "rotate", "coarse", "select", "prep", "scan",
 * "rerank", "sort", "replay", "total".  Also the algorithmic bytes the scan launches covered
 * (sum over probed lists of len * (dim/8 + 16)) and the number of scan launches. */
typedef struct {
    uint32_t struct_size;      /* in: sizeof(rq_profile_t) of the caller */
    uint32_t coarse_fallback_rows; /* queries whose pre-filtered coarse ranking (option "coarse_impl") fell back to all-lists exact order */
    float ms_rotate, ms_coarse, ms_select, ms_prep, ms_group, ms_scan, ms_rerank, ms_sort, ms_replay,
        ms_total;
    uint64_t scan_bytes;       /* algorithmic bytes over all scan launches of the call */
    uint64_t scan_candidates;  /* (query, candidate) pairs scanned                      */
    uint64_t rerank_candidates;/* accurate distances computed (superset of `precise`)   */
    uint32_t scan_launches;
    uint32_t retries;          /* queries re-run because a survivor buffer overflowed   */
    /* the matrix-core launches alone (a subset of the scan figures above) */
    float ms_scan_matrix;      /* device time of the scan_mfma_kernel launches           */
    uint32_t matrix_launches;
    uint64_t matrix_pairs;     /* (query, candidate) pairs they scored                   */
    /* 32x32 (query x candidate) sub-tile steps of the matrix-core scan, and how many of them were flagged by its gate and took
     * the exact f32 path (always filled since ABI revision 4) */
    uint64_t matrix_subtile_steps, matrix_exact_steps;
    /* of rerank_candidates: survivors the shadow rows (8-bit or fp16, option "rerank_shadow") proved to be at or above their stage's threshold, whose
     * 4*dim-byte row was therefore never fetched (option "rerank_shadow"; large batches) */
    uint64_t rerank_shadow_rejects;
    /* small-batch path (<= 64 queries: rotate + coarse in one launch, then one block per query doing probe selection, query
     * quantisation and the early stages in LDS): device time of that second launch, and how many passes took the path */
    float ms_early;
    uint32_t small_batch_passes;
    /* survivor records + run directories (+ the shared arena) of the call's largest pass, bytes; passes in which a stage
     * appended its survivors to the shared arena and scattered them into per-query segments sized by their exact counts,
     * instead of one capacity for every query */
    uint64_t survivor_workspace_bytes;
    uint32_t segmented_passes;
    uint32_t matrix_additive_launches; /* of matrix_launches: those that ran the additive gate (no threshold MFMA; option "scan_gate") */
} rq_profile_t;
/* level: 0 = off; 1 = every kernel group bracketed (each event costs a few microseconds of stream
 * time); 2 = only the scan launches and the whole pass (ms_scan, ms_total; the other fields stay 0). */
rq_status rq_set_profiling(int level);
/* Engine options.  "scan_impl": 0 = auto (default: fp6 matrix-core scan when many queries share each
 * list, v_dot8 VALU scan otherwise), 1 = VALU only, 2 = matrix cores wherever available.  All
 * settings return identical results; the option exists for tests and measurements.
 * "base_device_mb": HBM budget (MiB) of the raw vectors of indexes built / loaded from now on (-1 = automatic, the
 * default); vectors beyond it live in pinned host memory.  Results never depend on it.
 * "pass_overlap": a call that needs several passes (more than 65 536 queries) keeps two of them in flight, each on a workspace
 * and stream of its own (1, default), or runs them one after the other (0).  Results are identical.
 * "split_rows": indexes built / loaded from now on whose raw vectors leave no room for shadow rows -- tiered ones (both tiers),
 * and untiered ones where the shadow of "rerank_shadow" would not fit -- keep each raw vector as two 16-bit planes inside its own
 * 4*dim bytes (1, default: the upper halves of the f32 words rounded to nearest, then the lower halves; every word is restored
 * exactly), or as plain f32 (0); 2 = split rows always (test hook).  The re-ranker of large batches reads the first plane alone
 * and fetches the second one only for survivors the first cannot prove out of the top-k.  Such an index has no f32 device array
 * of its raw vectors (rq_get_device_ptr(RQ_ARR_BASE) -> RQ_ERR_UNSUPPORTED; rq_get_array restores them).  Results are
 * bit-identical for every value.
 * "rerank_shadow": shadow rows of indexes built / loaded from now on whose raw vectors are all in HBM (when that still leaves
 * the query workspaces their room): 2 (default) = 8-bit rows (dim bytes per vector; one affine map per list, error bound
 * measured while the codes are written), 1 = fp16 rows (2*dim bytes per vector), 0 = none.  The re-ranker of large batches
 * reads the shadow row first and fetches the f32 row only when the shadow cannot prove that the exact distance is at or
 * above the stage's threshold (then the reference rejects it whatever its value).  Results are bit-identical for every value.
 * "max_scan_blocks": test hook, blocks per scan launch (0 = hardware bound).
 * "coarse_impl": coarse ranking (identical probe lists and distance bits for every value): 0 = automatic (default: batches of
 * >= 2048 queries rank through the bf16 matrix-core pre-filter + exact-order refinement of the candidates within a proven margin
 * of the nprobe-th approximate distance, where it applies -- dim/64 in {1, 2, 3, 4, 6, 8, 12}, <= 65536 lists, nprobe <= 64; the
 * nprobe-th approximate distance is bounded from the minima of 32-list tiles from 4096 lists up (dim <= 512) and always
 * beyond 8192 lists, with the row in one wave's registers otherwise --, else the exact-order distance kernels over all lists),
 * 1 = exact-order kernel with the query rows through LDS, 2 = exact-order kernel with the query rows in scalar registers,
 * 3 = the pre-filter wherever it applies, row in registers up to 8192 lists (tests), 4 = the pre-filter with the tile-minima
 * selection wherever a row has at least nprobe tiles (tests).  rq_profile_t.coarse_fallback_rows counts queries whose
 * candidate set exceeded the refinement's 256 slots (near-equidistant centroids) or whose margin was not finite: they are
 * ranked in exact order over all lists (per row).
 * "pair_split": test hook, 1 (default) = passes over an index most of whose lists are empty (a shard: the probe lists name the lists
 * of every shard) settle the (query, list) pairs with nothing to scan by one thread each and run the query quantisation and the final
 * stage's record fill over a compacted list of the others, 0 = every pair takes the full path.  Identical results.
 * "coarse_tiled_from": developer knob, list count from which the automatic choice selects through tile minima (default 4096).
 * "shared_thresholds": rq_query_batch_sharded_device: 1 (default) = with more than one shard the step runs the nearest
 * list first, all-reduces (min) the k-th best distances and seeds the rest of the probe list with them (see
 * rq_query_batch_device_seeded); 0 = every shard prunes with its own thresholds only; 2 = also with one shard (tests).
 * "survivor_segments": 1 (default) = once an index has shown that its batches overflow the default survivor capacity, large
 * batches (>= 256 queries) size the survivor buffers of the stages that can exceed it PER QUERY (the scan appends to one shared
 * arena while counting per query; exact counts + prefix sum + a scatter pass) instead of giving every query the worst one's
 * capacity; 0 = never, 2 = every large batch (tests).  A pass whose arena cannot be had -- no memory, or it keeps overflowing --
 * is repeated on the uniform buffers; the failure injection that exercises this (value 3: as 2 with every arena stage failing
 * through a real, oversized allocation) exists in the developer build only (`make dev`, -DRQ_DEV_ABLATIONS) and is refused
 * here.  Identical results.
 * "small_batch": 0 (default) = batches of <= 64 queries (the reference's one-query-per-call loop included) run as a handful
 * of fat launches (kernels_small.h) whenever the shape allows (nprobe <= 64, <= 8192 lists, topk <= 256, dim in {64, 128,
 * 256, 512, 768, 1024}), 1 = never (test hook).  Identical results.
 * "dense_dir": test hook, 1 (default) = the VALU stages of large batches write their survivor runs into a directory
 * indexed by stream position (nothing to sort), 0 = runs are appended and the directory is sorted.
 * "group_rank": test hook, placement of a cluster-major stage's (query, list) pairs: 0 = one atomic per pair,
 * 1 = automatic (default: per-block LDS histograms for big stages), 2 = histograms whenever they fit.
 * "assign_impl": nearest-list assignment of builds from now on: 0 (default) = bf16 matrix-core pre-filter + exact-order
 * refinement wherever the shape is supported, 1 = exact-order kernels only (test hook); bit-identical labels and distances.
 * "scan_tile_table": test hook, grids of the list-major scan stages: 0 = lists x tiles-of-the-longest-list everywhere,
 * 1 (default) = one block per existing (list, tile) when most of the plain grid would be empty blocks, 2 = always.
 * "small_batch_span": developer knob, stream positions a query's block of the small-batch path scans itself at most
 * (default 2560; results are identical for every value).
 * "scan_gate": gate of the matrix-core scan (results never depend on it): 0 (default) = the additive bound S* >= B_q + G_c
 * where it exists (dim 64 / 128, stages on the uniform survivor buffers: no threshold MFMA) until an index shows that it sends
 * more than 3 % of the sub-tile steps down the exact path, 1 = the bf16 rank-5 threshold MFMA always, 2 = the additive bound
 * wherever it exists (test hook).
 * Developer knobs (results are identical for every value): "stage_growth" (geometric growth of the early stages, 0 = default),
 * "large_batch_from" (queries from which a batch takes the large-batch form of the stages -- full-chip rerank / ordering / replay
 * launches, thin early stages, dense run directories, survivor arena; default 256), "cluster_major_div" (a VALU stage goes
 * list-major once its (query, list) pairs reach k / this; default 32), "stage_settle_pct" (where a large batch's early stages end,
 * in percent of the average list length; default 100), "scan_debug": measurement hooks under which every result is UNCHANGED -- 128 (kept for older hosts: the sub-tile /
 * exact-path step counters of rq_profile_t are always on since ABI revision 4), 512 no shadow rows in the rerank,
 * 4096 the phases of the small-batch kernel, 16384 the stage list of every pass (stderr).  Any other bit is refused with
 * RQ_ERR_INVALID by this library: the TIMING ABLATIONS of the matrix-core scan (1, 2, 4, 64, 1024, 8192: results are WRONG) and
 * its in-kernel cycle counters (256) are compiled only into the developer build (make -C rabitq_amd/csrc dev ->
 * librabitq_hip_dev.so, -DRQ_DEV_ABLATIONS; scripts/exp/ select it through RABITQ_HIP_SO), so the shipped kernels carry none
 * of those branches.
 *
 * Threading: options are process-global and result-neutral (every value of every option leaves ids, distances and counters
 * unchanged: tests/test_gpu_parity.py::test_every_option_value_keeps_golden_results).  A pass reads the ones that pick a
 * kernel or a record format once per stage, so changing an option while queries are in flight on other threads only
 * changes which kernels later stages / passes use (test_options_flipped_under_concurrent_queries).
 * Removed in ABI revision 3: "scan_dense", "coarse_impl" = 3 (both answer RQ_ERR_INVALID). */
rq_status rq_set_option(const char *name, int value);
rq_status rq_last_profile(rq_profile_t *out);

#ifdef __cplusplus
}
#endif
#endif
