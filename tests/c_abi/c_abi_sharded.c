/* Plain-C client of the multi-GPU entry points of the C ABI (include/rii_amd.h, round 4): a C / C++ caller shards the query path
 * without any Python -- rii_comm_unique_id / rii_comm_init (RCCL behind the library), rii_query_linear_qsharded_dev,
 * rii_query_ivf_qsharded_dev, rii_query_linear_dbsharded_dev, and the device-queries -> host-rows call
 * rii_query_linear_dev_to_host; round 5: L past 8192 candidates, shard offsets in the record headers, a rank-local failure.  GPU box only (world size 1: the collective still runs through RCCL); every row is compared with
 * the single-engine host-pointer calls. */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "rii_amd.h"

#define M 16
#define KS 256
#define DS 4
#define N 40000
#define B 70
#define D (M * DS)

#define CHECK(x) do { if ((x) != RII_OK) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, rii_last_error()); return 1; } } while (0)
#define HCHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "%s:%d hip error\n", __FILE__, __LINE__); return 1; } } while (0)

static int same_rows(const int64_t *a, const float *ad, const int64_t *b, const float *bd, int n, const char *what)
{
    for (int i = 0; i < n; ++i)
        if (a[i] != b[i] || memcmp(&ad[i], &bd[i], 4) != 0) {
            fprintf(stderr, "%s: row entry %d differs: (%lld, %g) vs (%lld, %g)\n", what, i, (long long) a[i], ad[i], (long long) b[i], bd[i]);
            return 0;
        }
    return 1;
}

int main(void)
{
    if (rii_device_count() <= 0) { printf("no GPU\n"); return 0; }
    float *cw = malloc(sizeof(float) * M * KS * DS), *q = malloc(sizeof(float) * B * D);
    uint8_t *codes = malloc((size_t) N * M);
    unsigned s = 777u;
    for (int i = 0; i < M * KS * DS; ++i) { s = s * 1664525u + 1013904223u; cw[i] = (float) ((s >> 8) % 256); }   /* integer tables: exact ties happen */
    for (int i = 0; i < B * D; ++i) { s = s * 1664525u + 1013904223u; q[i] = (float) ((s >> 8) % 256); }
    for (size_t i = 0; i < (size_t) N * M; ++i) { s = s * 1664525u + 1013904223u; codes[i] = (uint8_t) (s >> 13); }
    memcpy(codes + (size_t) 5000 * M, codes + (size_t) 100 * M, (size_t) 300 * M);                                  /* duplicated codes */

    rii_engine *e = NULL;
    CHECK(rii_create(cw, M, KS, DS, 0, RII_SIMD_AVX512, 0, &e));
    CHECK(rii_add_codes(e, codes, N, 0));
    CHECK(rii_reconfigure(e, 50, 3));

    unsigned char id[RII_COMM_ID_BYTES];
    rii_comm *c = NULL;
    CHECK(rii_comm_unique_id(id));
    CHECK(rii_comm_init(id, 0, 1, 0, &c));
    if (rii_comm_rank(c) != 0 || rii_comm_size(c) != 1) return 2;

    float *dq; int64_t *d_ids, *d_cnt; float *d_d; int32_t *d_tie;
    HCHECK(hipMalloc((void **) &dq, sizeof(float) * B * D));
    HCHECK(hipMalloc((void **) &d_ids, sizeof(int64_t) * B * 8));
    HCHECK(hipMalloc((void **) &d_d, sizeof(float) * B * 8));
    HCHECK(hipMalloc((void **) &d_cnt, sizeof(int64_t) * B));
    HCHECK(hipMalloc((void **) &d_tie, sizeof(int32_t) * B));
    HCHECK(hipMemcpy(dq, q, sizeof(float) * B * D, hipMemcpyHostToDevice));

    int64_t want_i[B * 8], got_i[B * 8], want_c[B], got_c[B];
    float want_d[B * 8], got_d[B * 8];
    for (int topk = 1; topk <= 5; topk += 4) {
        CHECK(rii_query_linear(e, q, B, topk, NULL, 0, want_i, want_d));
        /* query-sharded */
        CHECK(rii_query_linear_qsharded_dev(e, c, dq, B, topk, NULL, 0, d_ids, d_d, NULL));
        CHECK(rii_synchronize(e));
        HCHECK(hipMemcpy(got_i, d_ids, sizeof(int64_t) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_d, d_d, sizeof(float) * B * topk, hipMemcpyDeviceToHost));
        if (!same_rows(got_i, got_d, want_i, want_d, B * topk, "qsharded linear")) return 3;
        /* database-sharded (one shard = the whole database at world size 1; exact ties replayed in std::partial_sort's order) */
        CHECK(rii_query_linear_dbsharded_dev(e, c, 0, dq, B, topk, NULL, 0, 0, d_ids, d_d, d_tie, NULL, 0, NULL));
        CHECK(rii_synchronize(e));
        HCHECK(hipMemcpy(got_i, d_ids, sizeof(int64_t) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_d, d_d, sizeof(float) * B * topk, hipMemcpyDeviceToHost));
        if (!same_rows(got_i, got_d, want_i, want_d, B * topk, "dbsharded linear")) return 4;
        /* device queries -> host rows */
        memset(got_i, 0xff, sizeof(got_i));
        CHECK(rii_query_linear_dev_to_host(e, dq, B, topk, NULL, 0, got_i, got_d, NULL));
        if (!same_rows(got_i, got_d, want_i, want_d, B * topk, "dev_to_host linear")) return 5;
        /* inverted index, query-sharded */
        CHECK(rii_query_ivf(e, q, B, topk, NULL, 0, 2000, want_i, want_d, want_c));
        CHECK(rii_query_ivf_qsharded_dev(e, c, dq, B, topk, NULL, 0, 2000, d_ids, d_d, d_cnt, NULL));
        CHECK(rii_synchronize(e));
        HCHECK(hipMemcpy(got_i, d_ids, sizeof(int64_t) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_d, d_d, sizeof(float) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_c, d_cnt, sizeof(int64_t) * B, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            if (got_c[b] != want_c[b]) { fprintf(stderr, "ivf count differs at %d\n", b); return 6; }
            if (!same_rows(got_i + b * topk, got_d + b * topk, want_i + b * topk, want_d + b * topk, (int) want_c[b], "qsharded ivf")) return 6;
        }
        /* inverted index, database-sharded (one shard = the whole database at world size 1: list lengths all-gathered, the global
         * walk replayed, rows merged by (distance, position), exact ties replayed from all candidates) */
        CHECK(rii_query_ivf_dbsharded_dev(e, c, 0, N, dq, B, topk, NULL, 0, 0, 2000, d_ids, d_d, d_cnt, d_tie, NULL));
        CHECK(rii_synchronize(e));
        HCHECK(hipMemcpy(got_i, d_ids, sizeof(int64_t) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_d, d_d, sizeof(float) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_c, d_cnt, sizeof(int64_t) * B, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            if (got_c[b] != want_c[b]) { fprintf(stderr, "dbsharded ivf count differs at %d\n", b); return 7; }
            if (!same_rows(got_i + b * topk, got_d + b * topk, want_i + b * topk, want_d + b * topk, (int) want_c[b], "dbsharded ivf")) return 7;
        }
    }
    /* round 5: the database-sharded inverted index past the 8192 candidates a launch used to sort (the reference's billion-scale run
     * asks for L = sqrt(N) ~ 31.6 k: examples/benchmark/run_sift1b.py:105-106), top-1 and top-5, exact ties replayed from sequences
     * rebuilt in global scratch */
    for (int topk = 1; topk <= 5; topk += 4) {
        const int64_t L = 30000;
        CHECK(rii_query_ivf(e, q, B, topk, NULL, 0, L, want_i, want_d, want_c));
        CHECK(rii_query_ivf_dbsharded_dev(e, c, 0, N, dq, B, topk, NULL, 0, 0, L, d_ids, d_d, d_cnt, d_tie, NULL));
        CHECK(rii_synchronize(e));
        HCHECK(hipMemcpy(got_i, d_ids, sizeof(int64_t) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_d, d_d, sizeof(float) * B * topk, hipMemcpyDeviceToHost));
        HCHECK(hipMemcpy(got_c, d_cnt, sizeof(int64_t) * B, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            if (got_c[b] != want_c[b]) { fprintf(stderr, "dbsharded ivf (L = 30000) count differs at %d\n", b); return 8; }
            if (!same_rows(got_i + b * topk, got_d + b * topk, want_i + b * topk, want_d + b * topk, (int) want_c[b], "dbsharded ivf L=30000")) return 8;
        }
    }
    /* the shard's first id travels in the record header: another placement of the same shard on the SAME communicator, then back */
    {
        const int64_t offs[3] = {123456789, 0, 77};
        CHECK(rii_query_linear(e, q, B, 1, NULL, 0, want_i, want_d));
        for (int t = 0; t < 3; ++t) {
            CHECK(rii_query_linear_dbsharded_dev(e, c, offs[t], dq, B, 1, NULL, 0, 0, d_ids, d_d, NULL, NULL, 0, NULL));
            CHECK(rii_synchronize(e));
            HCHECK(hipMemcpy(got_i, d_ids, sizeof(int64_t) * B, hipMemcpyDeviceToHost));
            for (int b = 0; b < B; ++b)
                if (got_i[b] != want_i[b] + offs[t]) { fprintf(stderr, "id offset %lld: row %d is %lld\n", (long long) offs[t], b, (long long) got_i[b]); return 9; }
        }
    }
    /* a rank-local failure (more local target ids than local codes) is this rank's error -- the call still took part in its
     * collective, so the communicator stays in step and the next call works */
    {
        int64_t *d_bad;
        HCHECK(hipMalloc((void **) &d_bad, sizeof(int64_t) * (N + 1)));
        HCHECK(hipMemset(d_bad, 0, sizeof(int64_t) * (N + 1)));
        if (rii_query_linear_dbsharded_dev(e, c, 0, dq, B, 1, d_bad, N + 1, 2 * (int64_t) N, d_ids, d_d, NULL, NULL, 0, NULL) == RII_OK) return 10;
        CHECK(rii_query_linear_dbsharded_dev(e, c, 0, dq, B, 1, NULL, 0, 0, d_ids, d_d, NULL, NULL, 0, NULL));
        CHECK(rii_synchronize(e));
        HCHECK(hipMemcpy(got_i, d_ids, sizeof(int64_t) * B, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            if (got_i[b] != want_i[b]) return 11;
        HCHECK(hipFree(d_bad));
    }
    rii_comm_destroy(c);
    rii_destroy(e);
    printf("C ABI sharded OK\n");
    return 0;
}
