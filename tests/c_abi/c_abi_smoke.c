/* Plain-C client of the C ABI (include/rii_amd.h): what a cgo / JNI / N-API / ctypes binding sees.
 * CPU box: must compile, link and fail loudly in rii_create (no GPU).  GPU box: builds an index, queries it and
 * cross-checks the batched linear result against a brute-force ADC written here in C (exact for this tiny case
 * because Ds = 1: one subtraction and one multiplication per table entry, summed in m order). */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "rii_amd.h"

#define M 4
#define KS 16
#define DS 1
#define N 500
#define B 3

int main(void)
{
    float cw[M * KS * DS], q[B * M * DS];
    uint8_t codes[N * M];
    unsigned s = 12345u;
    for (int i = 0; i < M * KS * DS; ++i) { s = s * 1664525u + 1013904223u; cw[i] = (float) ((s >> 8) % 1000) / 10.0f; }
    for (int i = 0; i < B * M * DS; ++i) { s = s * 1664525u + 1013904223u; q[i] = (float) ((s >> 8) % 1000) / 10.0f; }
    for (int i = 0; i < N * M; ++i) { s = s * 1664525u + 1013904223u; codes[i] = (uint8_t) ((s >> 8) % KS); }

    if (rii_device_count() <= 0) {
        rii_engine *e = NULL;
        int rc = rii_create(cw, M, KS, DS, 0, RII_SIMD_AVX512, 0, &e);
        if (rc != RII_ERR_HIP || e != NULL) { fprintf(stderr, "expected RII_ERR_HIP without a GPU, got %d\n", rc); return 2; }
        printf("no GPU: rii_create failed loudly: %s\n", rii_last_error());
        return 0;
    }
    rii_engine *e = NULL;
    if (rii_create(cw, M, KS, DS, 0, RII_SIMD_AVX512, 0, &e) != RII_OK) { fprintf(stderr, "%s\n", rii_last_error()); return 1; }
    if (rii_add_codes(e, codes, N, 0) != RII_OK) { fprintf(stderr, "%s\n", rii_last_error()); return 1; }
    if (rii_reconfigure(e, 10, 3) != RII_OK) { fprintf(stderr, "%s\n", rii_last_error()); return 1; }
    int64_t ids[B], cnt[B], ids_ivf[B];
    float d[B], d_ivf[B];
    if (rii_query_linear(e, q, B, 1, NULL, 0, ids, d) != RII_OK) { fprintf(stderr, "%s\n", rii_last_error()); return 1; }
    if (rii_query_ivf(e, q, B, 1, NULL, 0, N, ids_ivf, d_ivf, cnt) != RII_OK) { fprintf(stderr, "%s\n", rii_last_error()); return 1; }
    for (int b = 0; b < B; ++b) {
        float best = 1e30f; int64_t bi = -1;
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int m = 0; m < M; ++m) {
                volatile float t = q[b * M + m] - cw[m * KS + codes[n * M + m]];
                volatile float t2 = t * t;
                acc = acc + t2;
            }
            if (acc < best) { best = acc; bi = n; }
        }
        if (bi != ids[b] || best != d[b] || d_ivf[b] != d[b] || cnt[b] != 1) {
            fprintf(stderr, "mismatch b=%d: gpu (%lld, %g) ivf (%lld, %g) c (%lld, %g)\n", b, (long long) ids[b], d[b],
                    (long long) ids_ivf[b], d_ivf[b], (long long) bi, best);
            return 3;
        }
    }
    int rc = rii_add_codes(e, codes, 1, 1);
    if (rc != RII_OK) { fprintf(stderr, "%s\n", rii_last_error()); return 1; }
    if (rii_get_N(e) != N + 1 || rii_get_nlist(e) != 10) return 4;
    rii_destroy(e);
    printf("C ABI smoke OK\n");
    return 0;
}
