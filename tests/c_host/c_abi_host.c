/* A compiled host on the drop-in boundary: plain C, only include/rabitq_hip.h and librabitq_hip.so (what a Rust
 * `extern "C"` block would bind).  Builds an index from seeded vectors, dumps / reloads it through the
 * crate's directory format, and checks that per-vector rq_query equals rq_query_batch and the reloaded index.
 *   gcc -O2 -I include tests/c_host/c_abi_host.c -o c_abi_host -L rabitq_amd -lrabitq_hip -Wl,-rpath,$PWD/rabitq_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rabitq_hip.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        rq_status s_ = (call);                                                         \
        if (s_ != RQ_OK) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #call, (int)s_, rq_last_error());        \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

static unsigned long long rng_state = 88172645463325252ull;
static float frand(void) { /* xorshift64*, uniform in [-1, 1) */
    rng_state ^= rng_state >> 12, rng_state ^= rng_state << 25, rng_state ^= rng_state >> 27;
    return (float)((rng_state * 2685821657736338717ull) >> 40) / 8388608.0f - 1.0f;
}

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/tmp/rq_c_host_index";
    const unsigned n = 6000, d = 100, k = 12, nq = 40, probe = 5, topk = 7; /* d = 100 pads to 128 */
    float *base = malloc(sizeof(float) * n * d), *cent = malloc(sizeof(float) * k * d), *q = malloc(sizeof(float) * nq * d);
    for (unsigned j = 0; j < k * d; ++j) cent[j] = 2.0f * frand();
    for (unsigned i = 0; i < n; ++i)
        for (unsigned e = 0; e < d; ++e) base[i * d + e] = cent[(i % k) * d + e] + 0.6f * frand();
    for (unsigned i = 0; i < nq; ++i)
        for (unsigned e = 0; e < d; ++e) q[i * d + e] = cent[((i * 7) % k) * d + e] + 0.6f * frand();

    CHECK(rq_init(0));
    rq_index *idx = NULL, *idx2 = NULL;
    CHECK(rq_build(base, n, d, cent, k, NULL /* seeded Gaussian-QR rotation */, 12345, &idx));
    if (rq_abi_version() != RQ_ABI_VERSION) {
        fprintf(stderr, "library ABI %u, header ABI %u\n", rq_abi_version(), (unsigned)RQ_ABI_VERSION);
        return 3;
    }
    rq_info_t info;
    info.struct_size = sizeof info; /* sized out-struct: the library writes at most this many bytes */
    CHECK(rq_info(idx, &info));
    if (info.dim != 128 || info.k != k || info.n != n || info.n_hbm != n) {
        fprintf(stderr, "unexpected rq_info\n");
        return 3;
    }
    CHECK(rq_dump_dir(idx, dir));
    CHECK(rq_load_dir(dir, &idx2));

    float *bd = malloc(sizeof(float) * nq * topk), sd[16], ld[16];
    uint32_t *bi = malloc(sizeof(uint32_t) * nq * topk), *bn = malloc(sizeof(uint32_t) * nq), si[16], li[16], sn, ln;
    CHECK(rq_metrics_reset());
    CHECK(rq_query_batch(idx, q, nq, d, probe, topk, 0, bd, bi, bn));
    rq_metrics_t mb, ms;
    CHECK(rq_metrics(&mb));
    CHECK(rq_metrics_reset());
    int bad = 0;
    for (unsigned i = 0; i < nq; ++i) {
        CHECK(rq_query(idx, q + i * d, d, probe, topk, 0, sd, si, &sn));   /* RaBitQ::query, one vector per call */
        CHECK(rq_query(idx2, q + i * d, d, probe, topk, 0, ld, li, &ln));  /* the reloaded index */
        if (sn != bn[i] || ln != sn || memcmp(si, bi + i * topk, sn * 4) || memcmp(sd, bd + i * topk, sn * 4) ||
            memcmp(li, si, sn * 4) || memcmp(ld, sd, sn * 4))
            ++bad;
        /* returned distances are the squared L2 of the returned ids (tolerance: summation order only) */
        for (unsigned r = 0; r < sn; ++r) {
            double acc = 0;
            for (unsigned e = 0; e < d; ++e) {
                double t = (double)base[si[r] * d + e] - q[i * d + e];
                acc += t * t;
            }
            if (fabs(acc - sd[r]) > 1e-4 * acc) ++bad;
        }
    }
    CHECK(rq_metrics(&ms));
    /* two single-query loops = twice the batch's counters (src/metrics.rs) */
    if (ms.query != 2 * mb.query || ms.rough != 2 * mb.rough || ms.precise != 2 * mb.precise) ++bad;
    /* error behaviour: the reference asserts / panics, the C ABI returns a status */
    if (rq_query(idx, q, 200, probe, topk, 0, sd, si, &sn) != RQ_ERR_DIM_MISMATCH) ++bad;
    if (rq_query(idx, q, d, 0, topk, 0, sd, si, &sn) != RQ_ERR_INVALID) ++bad;
    rq_free(idx);
    rq_free(idx2);
    printf("c_abi_host: %u queries, batch == single == reloaded: %s (query %llu rough %llu precise %llu)\n", nq,
           bad ? "MISMATCH" : "ok", (unsigned long long)mb.query, (unsigned long long)mb.rough, (unsigned long long)mb.precise);
    return bad ? 1 : 0;
}
