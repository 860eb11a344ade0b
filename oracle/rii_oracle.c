/*
 * rii_oracle.c -- CPU restatement of the rii IVFPQ query hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker*, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (rii_amd/csrc) never links it.
 *
 * Parity status: PINNED.  Every function below is validated bit-for-bit in this container against the
 * real reference (matsui528/rii v0.2.12) compiled unmodified to oracle/_ref (see oracle/Makefile,
 * tests/gen_golden.py, tests/test_oracle_vs_ref.py) and against the committed fixtures tests/golden/*.npz
 * generated from that build.
 *
 * Each function cites the reference file:line it restates (paths relative to /root/reference).
 * Arithmetic notes that are NOT visible in the reference source but were read out of the -Ofast object
 * code (GCC 11.4, flags of setup.py:87-100) are marked [objcode].
 *
 * Build: `make -C oracle oracle` (plain C11, -ffp-contract=off so that every fmaf below is explicit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stddef.h>

#define ORACLE_API __attribute__((visibility("default")))

/* g_simd_architecture of src/distance.h:114,173,220: which fvec_L2sqr variant -march=native selected. */
enum { RII_SIMD_SSE = 0, RII_SIMD_AVX = 1, RII_SIMD_AVX512 = 2 };

/* ------------------------------------------------------------------------------------------------
 * a1. fvec_L2sqr  --  src/distance.h:117-170 (AVX512), :177-217 (AVX), :225-252 (SSE)
 *
 * Lane-structured accumulation: W-wide chunk accumulators folded hi+lo down to 4 lanes, a 4-wide chunk,
 * a zero-padded (masked_read, :44-105) tail, then two _mm_hadd_ps = (l0+l1)+(l2+l3).
 * [objcode] GCC -Ofast contracts every `msum = msum + d*d` of the AVX/AVX512 variants into one FMA
 * (those builds imply -mfma); the SSE variant (no AVX => no FMA) stays mul-then-add.
 * ---------------------------------------------------------------------------------------------- */
static inline float sq_acc(float acc, float d, int fused)
{
    return fused ? fmaf(d, d, acc) : acc + d * d;
}

ORACLE_API float oracle_fvec_l2sqr(const float *x, const float *y, size_t d, int arch)
{
    const int fused = (arch != RII_SIMD_SSE);
    float l16[16], l8[8], l4[4];
    int i;
    for (i = 0; i < 16; ++i) l16[i] = 0.f;
    if (arch == RII_SIMD_AVX512) {
        while (d >= 16) {                                  /* distance.h:121-128 */
            for (i = 0; i < 16; ++i) l16[i] = sq_acc(l16[i], x[i] - y[i], fused);
            x += 16; y += 16; d -= 16;
        }
    }
    for (i = 0; i < 8; ++i) l8[i] = l16[8 + i] + l16[i];    /* distance.h:130-132 (hi + lo) */
    if (arch == RII_SIMD_AVX512 || arch == RII_SIMD_AVX) {
        while (d >= 8) {                                   /* distance.h:134-142 / :181-188 */
            for (i = 0; i < 8; ++i) l8[i] = sq_acc(l8[i], x[i] - y[i], fused);
            x += 8; y += 8; d -= 8;
        }
    }
    for (i = 0; i < 4; ++i) l4[i] = l8[4 + i] + l8[i];      /* distance.h:144-146 / :190-192 */
    if (arch == RII_SIMD_SSE) {
        while (d >= 4) {                                   /* distance.h:229-236 */
            for (i = 0; i < 4; ++i) l4[i] = sq_acc(l4[i], x[i] - y[i], fused);
            x += 4; y += 4; d -= 4;
        }
    } else if (d >= 4) {                                   /* distance.h:148-156 / :194-202 */
        for (i = 0; i < 4; ++i) l4[i] = sq_acc(l4[i], x[i] - y[i], fused);
        x += 4; y += 4; d -= 4;
    }
    if (d > 0) {                                           /* masked tail, distance.h:158-165 */
        for (i = 0; i < 4; ++i) {
            float t = (i < (int) d) ? x[i] - y[i] : 0.f;
            l4[i] = sq_acc(l4[i], t, fused);
        }
    }
    return (l4[0] + l4[1]) + (l4[2] + l4[3]);              /* two hadd, distance.h:167-169 */
}

/* a2. RiiCpp::DTable -- src/rii.h:361-373.  codewords (M,Ks,Ds) row-major; out (M*Ks) row-major. */
ORACLE_API void oracle_dtable(const float *codewords, int M, int Ks, int Ds, const float *q, int arch,
                              float *out)
{
    for (int m = 0; m < M; ++m)
        for (int ks = 0; ks < Ks; ++ks)
            out[(size_t) m * Ks + ks] =
                oracle_fvec_l2sqr(q + (size_t) m * Ds, codewords + ((size_t) m * Ks + ks) * Ds, (size_t) Ds, arch);
}

/* a3/a4. RiiCpp::ADist -- src/rii.h:375-394: strictly sequential fp32 sum over m = 0..M-1. */
ORACLE_API float oracle_adist(const float *dtable, int M, int Ks, const uint8_t *code)
{
    float dist = 0.f;
    for (int m = 0; m < M; ++m) dist += dtable[(size_t) m * Ks + code[m]];
    return dist;
}

/* ------------------------------------------------------------------------------------------------
 * std::partial_sort as shipped by libstdc++ (GCC 11: bits/stl_algo.h __partial_sort/__heap_select,
 * bits/stl_heap.h __make_heap/__pop_heap/__adjust_heap/__push_heap/__sort_heap), restated for
 * pair<size_t,float> compared on .second only -- the comparator of src/rii.h:234-235,279-280,312-313.
 * Restating the exact algorithm (not just "k smallest") reproduces the reference's behaviour on exactly
 * tied distances and the order of the elements *past* `middle`, which QueryIvf walks (src/rii.h:283-326).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t id; float dist; } oracle_pair;

static void adjust_heap(oracle_pair *first, ptrdiff_t hole, ptrdiff_t len, oracle_pair value)
{
    const ptrdiff_t top = hole;
    ptrdiff_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (first[child].dist < first[child - 1].dist) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    /* __push_heap */
    ptrdiff_t parent = (hole - 1) / 2;
    while (hole > top && first[parent].dist < value.dist) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

ORACLE_API void oracle_partial_sort(oracle_pair *first, size_t middle, size_t n)
{
    ptrdiff_t len = (ptrdiff_t) middle;
    if (len >= 2) {                                        /* __make_heap */
        ptrdiff_t parent = (len - 2) / 2;
        for (;;) {
            oracle_pair v = first[parent];
            adjust_heap(first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    if (len > 0) {
        for (size_t i = middle; i < n; ++i) {              /* __heap_select */
            if (first[i].dist < first[0].dist) {           /* __pop_heap(first, middle, i) */
                oracle_pair v = first[i];
                first[i] = first[0];
                adjust_heap(first, 0, len, v);
            }
        }
    }
    while (len > 1) {                                      /* __sort_heap */
        --len;
        oracle_pair v = first[len];
        first[len] = first[0];
        adjust_heap(first, 0, len, v);
    }
}

/* ------------------------------------------------------------------------------------------------
 * a5. RiiCpp::QueryLinear -- src/rii.h:195-242.
 * Returns the number of results written (== topk).  tids: sorted int64, S == 0 means "all".
 * nthreads > 1 restates the `#pragma omp parallel for` of rii.h:213,222 (results are order-independent).
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API int64_t oracle_query_linear(const float *codewords, int M, int Ks, int Ds, const uint8_t *codes,
                                       int64_t N, const float *q, int topk, const int64_t *tids, int64_t S,
                                       int arch, int64_t *out_ids, float *out_dists)
{
    float *dt = (float *) malloc(sizeof(float) * (size_t) M * Ks);
    oracle_dtable(codewords, M, Ks, Ds, q, arch, dt);                      /* rii.h:205 */
    int64_t cnt = (S == 0) ? N : S;
    oracle_pair *scores = (oracle_pair *) malloc(sizeof(oracle_pair) * (size_t) (cnt > 0 ? cnt : 1));
    if (S == 0) {
#pragma omp parallel for
        for (int64_t n = 0; n < N; ++n) {                                  /* rii.h:210-217 */
            scores[n].id = (uint64_t) n;
            scores[n].dist = oracle_adist(dt, M, Ks, codes + (size_t) n * M);
        }
    } else {
#pragma omp parallel for
        for (int64_t s = 0; s < S; ++s) {                                  /* rii.h:218-228 */
            scores[s].id = (uint64_t) tids[s];
            scores[s].dist = oracle_adist(dt, M, Ks, codes + (size_t) tids[s] * M);
        }
    }
    oracle_partial_sort(scores, (size_t) topk, (size_t) cnt);              /* rii.h:234-235 */
    for (int k = 0; k < topk; ++k) { out_ids[k] = (int64_t) scores[k].id; out_dists[k] = scores[k].dist; }
    free(scores); free(dt);
    return topk;
}

static int tid_binary_search(const int64_t *tids, int64_t S, int64_t v)     /* std::binary_search */
{
    int64_t lo = 0, hi = S;
    while (lo < hi) { int64_t mid = lo + (hi - lo) / 2; if (tids[mid] < v) lo = mid + 1; else hi = mid; }
    return lo < S && tids[lo] == v;
}

/* ------------------------------------------------------------------------------------------------
 * a6/a7. RiiCpp::QueryIvf -- src/rii.h:244-326.
 * Posting lists in CSR form: list `no` = pl_ids[pl_off[no] .. pl_off[no+1]).
 * Returns the number of results (topk, or 0 for the "vectors not found" return of rii.h:324-325).
 * ---------------------------------------------------------------------------------------------- */
ORACLE_API int64_t oracle_query_ivf(const float *codewords, int M, int Ks, int Ds, const uint8_t *codes,
                                    int64_t N, const uint8_t *coarse_centers, int64_t nlist,
                                    const int64_t *pl_off, const int32_t *pl_ids, const float *q, int topk,
                                    const int64_t *tids, int64_t S, int64_t L, int arch, int64_t *out_ids,
                                    float *out_dists)
{
    float *dt = (float *) malloc(sizeof(float) * (size_t) M * Ks);
    oracle_dtable(codewords, M, Ks, Ds, q, arch, dt);                      /* rii.h:256 */
    oracle_pair *coarse = (oracle_pair *) malloc(sizeof(oracle_pair) * (size_t) nlist);
    for (int64_t no = 0; no < nlist; ++no) {                               /* rii.h:259-264 */
        coarse[no].id = (uint64_t) no;
        coarse[no].dist = oracle_adist(dt, M, Ks, coarse_centers + (size_t) no * M);
    }
    size_t w;                                                              /* rii.h:266-277 */
    if (S == 0) w = (size_t) round((double) L * (double) nlist / (double) N);
    else        w = (size_t) round((double) L * (double) nlist / (double) S);
    w += 3;
    if ((size_t) nlist < w) w = (size_t) nlist;
    oracle_partial_sort(coarse, w, (size_t) nlist);                        /* rii.h:279-280 */

    oracle_pair *scores = (oracle_pair *) malloc(sizeof(oracle_pair) * (size_t) (L > 0 ? L : 1));
    size_t nsc = 0;
    int64_t result = 0;
    int coarse_cnt = 0, finished = 0;
    for (int64_t c = 0; c < nlist && !finished; ++c) {                     /* rii.h:286-321 */
        int64_t no = (int64_t) coarse[c].id;
        coarse_cnt++;
        for (int64_t p = pl_off[no]; p < pl_off[no + 1]; ++p) {
            int64_t n = pl_ids[p];
            if (S != 0 && !tid_binary_search(tids, S, n)) continue;        /* rii.h:294-296 */
            scores[nsc].id = (uint64_t) n;
            scores[nsc].dist = oracle_adist(dt, M, Ks, codes + (size_t) n * M);
            nsc++;
            if (nsc == (size_t) L) { finished = 1; break; }                /* rii.h:302-304 */
        }
        if (!finished && (size_t) coarse_cnt == w && nsc >= (size_t) topk) finished = 1;  /* rii.h:309 */
    }
    if (finished) {
        oracle_partial_sort(scores, (size_t) topk, nsc);                   /* rii.h:312-313 */
        for (int k = 0; k < topk; ++k) { out_ids[k] = (int64_t) scores[k].id; out_dists[k] = scores[k].dist; }
        result = topk;
    }
    free(scores); free(coarse); free(dt);
    return result;
}

/* ------------------------------------------------------------------------------------------------
 * a8. PQk-means symmetric tables -- src/pqkmeans.cpp:23-34 with L2SquaredDistance :164-173.
 * [objcode] `dist += (a-b)*(a-b)` over Ds is auto-vectorised by GCC -Ofast into:
 *   - a W-wide FMA main loop when Ds >= W, reduced hi+lo -> 4 lanes -> (l2+l0),(l3+l1) -> sum,
 *   - one W/2-wide *unfused* (mul, then tree-add) block when the remainder >= W/2, added to the scalar,
 *   - a scalar FMA chain for the rest;  W = 16 for the AVX512 build, 8 for the AVX2 build.
 * For Ds=4 this gives fma(d3,d3,fma(d2,d2,fma(d1,d1,d0*d0))) under AVX512 but (d1^2+d3^2)+(d0^2+d2^2)
 * under AVX2 -- the reference's tables really are build-dependent.  (SSE builds: treated as the AVX2
 * shape with W=4 and no FMA; not validated, no such build of the reference was exercised.)
 * ---------------------------------------------------------------------------------------------- */
static float hsum_tree(const float *v, int w)   /* w in {16,8,4}: fold hi+lo to 4, then movhlps/shufps adds */
{
    float t[16];
    for (int i = 0; i < w; ++i) t[i] = v[i];
    while (w > 4) { w /= 2; for (int i = 0; i < w; ++i) t[i] = t[w + i] + t[i]; }
    float a = t[2] + t[0], b = t[3] + t[1];
    return b + a;
}

ORACLE_API float oracle_l2sq_pqkmeans(const float *a, const float *b, int n, int arch)
{
    const int W = (arch == RII_SIMD_AVX512) ? 16 : (arch == RII_SIMD_AVX ? 8 : 4);
    const int fused = (arch != RII_SIMD_SSE);
    int i = 0;
    float acc = 0.f;
    if (n >= W) {
        float lanes[16];
        for (int l = 0; l < W; ++l) lanes[l] = 0.f;
        for (; i + W <= n; i += W)
            for (int l = 0; l < W; ++l) lanes[l] = sq_acc(lanes[l], a[i + l] - b[i + l], fused);
        acc = hsum_tree(lanes, W);
    }
    const int H = W / 2;
    if (H >= 4 && n - i >= H) {
        float sq[8];
        for (int l = 0; l < H; ++l) { float d = a[i + l] - b[i + l]; sq[l] = d * d; }
        acc = acc + hsum_tree(sq, H);
        i += H;
    }
    for (; i < n; ++i) acc = sq_acc(acc, a[i] - b[i], fused);
    return acc;
}

/* D[m][k1][k2], (M,Ks,Ks) row-major. */
ORACLE_API void oracle_symmetric_tables(const float *codewords, int M, int Ks, int Ds, int arch, float *D)
{
    for (int m = 0; m < M; ++m)
        for (int k1 = 0; k1 < Ks; ++k1)
            for (int k2 = 0; k2 < Ks; ++k2)
                D[((size_t) m * Ks + k1) * Ks + k2] =
                    oracle_l2sq_pqkmeans(codewords + ((size_t) m * Ks + k1) * Ds,
                                         codewords + ((size_t) m * Ks + k2) * Ds, Ds, arch);
}

/* PQKMeans::SymmetricDistance src/pqkmeans.cpp:152-162 + FindNearetCenterLinear :193-218
 * (sequential fp32 sum over m; argmin with strict `<` => first minimum wins). */
static int64_t nearest_center(const float *D, int M, int Ks, const uint8_t *code, const uint8_t *centers,
                              int64_t K, float *out_dist)
{
    float min_dist = FLT_MAX;
    int64_t min_i = -1;
    for (int64_t c = 0; c < K; ++c) {
        float dist = 0.f;
        for (int m = 0; m < M; ++m)
            dist += D[((size_t) m * Ks + code[m]) * Ks + centers[(size_t) c * M + m]];
        if (dist < min_dist) { min_i = c; min_dist = dist; }
    }
    if (out_dist) *out_dist = min_dist;
    return min_i;
}

/* a8. coarse assignment of RiiCpp::UpdatePostingLists -- src/rii.h:335-359 (predict_one per code). */
ORACLE_API void oracle_assign(const float *D, int M, int Ks, const uint8_t *codes, int64_t num,
                              const uint8_t *centers, int64_t nlist, int32_t *assign)
{
#pragma omp parallel for
    for (int64_t n = 0; n < num; ++n)
        assign[n] = (int32_t) nearest_center(D, M, Ks, codes + (size_t) n * M, centers, nlist, NULL);
}

/* ------------------------------------------------------------------------------------------------
 * f1. Reconfigure sampling + PQk-means fit -- src/rii.h:108-156, src/pqkmeans.cpp:46-133,177-191,223-260.
 * Needs libstdc++'s std::shuffle / uniform_int_distribution / minstd_rand0 / mt19937 (GCC 11), restated:
 *   - std::default_random_engine == minstd_rand0: x <- 16807 x mod (2^31-1), range [1, 2^31-2];
 *   - uniform_int_distribution downscaling: Lemire multiply-shift when the engine range is exactly
 *     2^32-1 (mt19937), else the classic scaling/rejection loop (bits/uniform_int_dist.h);
 *   - std::shuffle draws two swap positions from one variate while (range+1)^2 fits in the engine range
 *     (bits/stl_algo.h __gen_two_uniform_ints), else one per element.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t *s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t) i;
    s->idx = 624;
}
static uint32_t mt_next(mt19937_t *s)
{
    if (s->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

typedef struct { int kind; uint64_t lcg; mt19937_t mt; } urng_t;   /* kind 0: minstd_rand0, 1: mt19937 */
static uint64_t urng_min(const urng_t *g) { return g->kind == 0 ? 1u : 0u; }
static uint64_t urng_max(const urng_t *g) { return g->kind == 0 ? 2147483646ull : 4294967295ull; }
static uint64_t urng_next(urng_t *g)
{
    if (g->kind == 0) { g->lcg = (g->lcg * 16807ull) % 2147483647ull; return g->lcg; }
    return mt_next(&g->mt);
}

/* uniform_int_distribution<unsigned long>(0, urange)(g), urange < engine range (always true here) */
static uint64_t uniform_upto(urng_t *g, uint64_t urange)
{
    const uint64_t urngrange = urng_max(g) - urng_min(g);
    const uint64_t uerange = urange + 1;
    if (urngrange == 0xFFFFFFFFull) {                        /* _S_nd<uint64_t>(g, uint32_t range) */
        uint32_t range = (uint32_t) uerange;
        uint64_t product = (uint64_t) urng_next(g) * (uint64_t) range;
        uint32_t low = (uint32_t) product;
        if (low < range) {
            uint32_t threshold = (uint32_t) (-range) % range;
            while (low < threshold) { product = (uint64_t) urng_next(g) * (uint64_t) range; low = (uint32_t) product; }
        }
        return product >> 32;
    }
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    uint64_t ret;
    do { ret = urng_next(g) - urng_min(g); } while (ret >= past);
    return ret / scaling;
}

static void std_shuffle_u64(uint64_t *v, uint64_t n, urng_t *g)
{
    if (n == 0) return;
    const uint64_t urngrange = urng_max(g) - urng_min(g);
    const uint64_t urange = n;
    if (urngrange / urange >= urange) {
        uint64_t i = 1;
        if ((urange % 2) == 0) {
            uint64_t j = uniform_upto(g, 1);
            uint64_t t = v[i]; v[i] = v[j]; v[j] = t; ++i;
        }
        while (i != n) {
            const uint64_t swap_range = i + 1;
            /* __gen_two_uniform_ints(swap_range, swap_range + 1, g) */
            uint64_t x = uniform_upto(g, swap_range * (swap_range + 1) - 1);
            uint64_t p1 = x / (swap_range + 1), p2 = x % (swap_range + 1);
            uint64_t t = v[i]; v[i] = v[p1]; v[p1] = t; ++i;
            t = v[i]; v[i] = v[p2]; v[p2] = t; ++i;
        }
        return;
    }
    for (uint64_t i = 1; i < n; ++i) {
        uint64_t j = uniform_upto(g, i);
        uint64_t t = v[i]; v[i] = v[j]; v[j] = t;
    }
}

/* ids sampled by RiiCpp::Reconfigure, src/rii.h:113-124.  out must hold min(N, 100*nlist) entries. */
ORACLE_API int64_t oracle_reconfigure_sample(int64_t N, int64_t nlist, int64_t *out)
{
    int64_t len = N < nlist * 100 ? N : nlist * 100;
    uint64_t *ids = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) N);
    for (int64_t i = 0; i < N; ++i) ids[i] = (uint64_t) i;
    urng_t g; g.kind = 0; g.lcg = 123 % 2147483647ull;        /* default_random_engine(123) */
    std_shuffle_u64(ids, (uint64_t) N, &g);
    for (int64_t i = 0; i < len; ++i) out[i] = (int64_t) ids[i];
    free(ids);
    return len;
}

/* PQKMeans::fit, src/pqkmeans.cpp:46-133 on `data` (n codes).  centers (K,M) out, assignments optional. */
ORACLE_API void oracle_pqkmeans_fit(const float *D, int M, int Ks, const uint8_t *data, int64_t n, int64_t K,
                                    int iter, uint8_t *centers, int32_t *assignments)
{
    /* InitializeCentersByRandomPicking, pqkmeans.cpp:177-191: shuffle(iota(n), mt19937(0)), first K */
    uint64_t *ids = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) n);
    for (int64_t i = 0; i < n; ++i) ids[i] = (uint64_t) i;
    urng_t g; g.kind = 1; mt_seed(&g.mt, 0u);
    /* NB: the reference shuffles a vector<int>; the permutation is element-type independent. */
    std_shuffle_u64(ids, (uint64_t) n, &g);
    for (int64_t k = 0; k < K; ++k) memcpy(centers + (size_t) k * M, data + (size_t) ids[k] * M, (size_t) M);
    free(ids);

    int32_t *assign = (int32_t *) malloc(sizeof(int32_t) * (size_t) (n > 0 ? n : 1));
    uint8_t *old = (uint8_t *) malloc((size_t) K * M);
    int *hist = (int *) malloc(sizeof(int) * (size_t) Ks);
    float *vote = (float *) malloc(sizeof(float) * (size_t) Ks);
    for (int itr = 0; itr < iter; ++itr) {
        memcpy(old, centers, (size_t) K * M);                               /* centers_old = centers_new */
#pragma omp parallel for
        for (int64_t i = 0; i < n; ++i)                                    /* pqkmeans.cpp:88-94 */
            assign[i] = (int32_t) nearest_center(D, M, Ks, data + (size_t) i * M, old, K, NULL);
        if (itr != iter - 1) {                                             /* pqkmeans.cpp:109-123 */
            for (int64_t k = 0; k < K; ++k) {
                int64_t cnt = 0;
                for (int64_t i = 0; i < n; ++i) cnt += (assign[i] == k);
                if (cnt == 0) continue;                                    /* keep old centre */
                for (int m = 0; m < M; ++m) {                              /* ComputeCenterBySparseVoting :223-260 */
                    memset(hist, 0, sizeof(int) * (size_t) Ks);
                    for (int64_t i = 0; i < n; ++i)
                        if (assign[i] == k) hist[data[(size_t) i * M + m]]++;
                    for (int k2 = 0; k2 < Ks; ++k2) vote[k2] = 0.f;
                    for (int k1 = 0; k1 < Ks; ++k1) {
                        int freq = hist[k1];
                        if (freq == 0) continue;
                        /* [objcode] vote[k2] += (float)freq * D is contracted to one FMA per k2 */
                        for (int k2 = 0; k2 < Ks; ++k2)
                            vote[k2] = fmaf((float) freq, D[((size_t) m * Ks + k1) * Ks + k2], vote[k2]);
                    }
                    float min_dist = FLT_MAX; int min_ks = -1;
                    for (int ks = 0; ks < Ks; ++ks)
                        if (vote[ks] < min_dist) { min_ks = ks; min_dist = vote[ks]; }
                    centers[(size_t) k * M + m] = (uint8_t) min_ks;
                }
            }
        }
    }
    if (assignments) memcpy(assignments, assign, sizeof(int32_t) * (size_t) n);
    free(vote); free(hist); free(old); free(assign);
}

ORACLE_API const char *oracle_version(void) { return "rii_oracle 0.1 (restates matsui528/rii v0.2.12 hot path)"; }
