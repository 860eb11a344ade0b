// rii_device.h -- device-side helpers shared by the HIP translation units (exact reference arithmetic).
#pragma once
#include "rii_internal.h"

namespace riiamd {

#define RII_SIMD_SSE 0
#define RII_SIMD_AVX 1
#define RII_SIMD_AVX512 2

// ===================================================================================================
// exact arithmetic helpers
// ===================================================================================================
__device__ __forceinline__ float sq_acc(float acc, float d, bool fused)
{
    return fused ? __fmaf_rn(d, d, acc) : __fadd_rn(acc, __fmul_rn(d, d));
}

// fvec_L2sqr, src/distance.h:117-252, all three compile-time variants (see oracle/rii_oracle.c for the
// derivation of the lane order and FMA contraction).
__device__ __forceinline__ float fvec_l2sqr_body(const float *__restrict__ x, const float *__restrict__ y, int d, int arch)
{
    const bool fused = (arch != RII_SIMD_SSE);
    float l16[16], l8[8], l4[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) l16[i] = 0.f;
    if (arch == RII_SIMD_AVX512) {
        while (d >= 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) l16[i] = sq_acc(l16[i], __fsub_rn(x[i], y[i]), fused);
            x += 16; y += 16; d -= 16;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) l8[i] = __fadd_rn(l16[8 + i], l16[i]);
    if (arch == RII_SIMD_AVX512 || arch == RII_SIMD_AVX) {
        while (d >= 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) l8[i] = sq_acc(l8[i], __fsub_rn(x[i], y[i]), fused);
            x += 8; y += 8; d -= 8;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) l4[i] = __fadd_rn(l8[4 + i], l8[i]);
    if (arch == RII_SIMD_SSE) {
        while (d >= 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) l4[i] = sq_acc(l4[i], __fsub_rn(x[i], y[i]), fused);
            x += 4; y += 4; d -= 4;
        }
    } else if (d >= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[i] = sq_acc(l4[i], __fsub_rn(x[i], y[i]), fused);
        x += 4; y += 4; d -= 4;
    }
    if (d > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = (i < d) ? __fsub_rn(x[i], y[i]) : 0.f;
            l4[i] = sq_acc(l4[i], t, fused);
        }
    }
    return __fadd_rn(__fadd_rn(l4[0], l4[1]), __fadd_rn(l4[2], l4[3]));
}
__device__ inline float fvec_l2sqr_dev(const float *__restrict__ x, const float *__restrict__ y, int d, int arch)
{
    return fvec_l2sqr_body(x, y, d, arch);
}
// the same operations for a compile-time dimension with both vectors in registers (the loops fold; no memory access)
template <int D>
__device__ __forceinline__ float fvec_l2sqr_regs(const float (&x)[D], const float (&y)[D], int arch)
{
    // (one fully folded copy per SIMD variant: with a run-time variant the pointer steps of the body are conditional and the
    //  register arrays would be demoted to scratch)
    if (arch == RII_SIMD_AVX512) return fvec_l2sqr_body(x, y, D, RII_SIMD_AVX512);
    if (arch == RII_SIMD_AVX) return fvec_l2sqr_body(x, y, D, RII_SIMD_AVX);
    return fvec_l2sqr_body(x, y, D, RII_SIMD_SSE);
}

// Ds == 4 (the SIFT shape: D=128, M=32) in straight-line form with two 16-byte loads.  Identical for all three SIMD
// variants: one 4-wide chunk into zeroed lanes (fma(t,t,+0) == the rounded square == SSE's mul-then-add-to-zero), no
// tail, then (l0+l1)+(l2+l3) (distance.h:148-169 / :194-216 / :229-251).  Callers guarantee 16-byte alignment.
__device__ __forceinline__ float fvec_l2sqr_ds4v(const float4 &a, const float4 &c)
{
    const float t0 = __fsub_rn(a.x, c.x), t1 = __fsub_rn(a.y, c.y), t2 = __fsub_rn(a.z, c.z), t3 = __fsub_rn(a.w, c.w);
    return __fadd_rn(__fadd_rn(__fmul_rn(t0, t0), __fmul_rn(t1, t1)), __fadd_rn(__fmul_rn(t2, t2), __fmul_rn(t3, t3)));
}
// Ds == 2 (D = 128 at M = 64, the reference's own benchmark shape): one partial chunk into zeroed lanes, lanes 2 and 3 stay +0:
// (t0^2 + t1^2) + (0 + 0), the same for all three SIMD variants
__device__ __forceinline__ float fvec_l2sqr_ds2v(const float2 &a, const float2 &c)
{
    const float t0 = __fsub_rn(a.x, c.x), t1 = __fsub_rn(a.y, c.y);
    return __fadd_rn(__fadd_rn(__fmul_rn(t0, t0), __fmul_rn(t1, t1)), 0.0f);
}
__device__ __forceinline__ float fvec_l2sqr_vec(const float4 &a, const float4 &c);
__device__ __forceinline__ float fvec_l2sqr_vec(const float2 &a, const float2 &c) { return fvec_l2sqr_ds2v(a, c); }
__device__ __forceinline__ float fvec_l2sqr_ds4(const float *__restrict__ x, const float *__restrict__ y)
{
    const float4 a = *reinterpret_cast<const float4 *>(x);
    const float4 c = *reinterpret_cast<const float4 *>(y);
    const float t0 = __fsub_rn(a.x, c.x), t1 = __fsub_rn(a.y, c.y), t2 = __fsub_rn(a.z, c.z), t3 = __fsub_rn(a.w, c.w);
    return __fadd_rn(__fadd_rn(__fmul_rn(t0, t0), __fmul_rn(t1, t1)), __fadd_rn(__fmul_rn(t2, t2), __fmul_rn(t3, t3)));
}

__device__ __forceinline__ float fvec_l2sqr_vec(const float4 &a, const float4 &c) { return fvec_l2sqr_ds4v(a, c); }

// dispatcher used by the table kernels (x = query sub-vector at m*Ds floats, y = codeword at i*Ds floats: both 16-byte
// aligned when Ds == 4 because the query rows and the codeword array are)
__device__ __forceinline__ float fvec_l2sqr_any(const float *__restrict__ x, const float *__restrict__ y, int d, int arch)
{
    if (d == 4) return fvec_l2sqr_ds4(x, y);
    return fvec_l2sqr_dev(x, y, d, arch);
}

// The in-kernel table for an even Ds other than 4 at Ks = 256 (thread = ks): the codewords of U subspaces requested together
// (8-byte loads), then fvec_L2sqr's operations on registers.  The plain loop fetched one subspace at a time -- a chain of M dependent
// round trips, 8-9 us of latency per block at the Deep1B shape (M = 16, Ds = 6).
template <int DS, int U, int ARCH, int MC>
__device__ __forceinline__ void table_rows_regs_arch(float *__restrict__ lds, const float *__restrict__ q, const float *__restrict__ codewords, int Mrt, int tid)
{
    static_assert(DS % 2 == 0, "8-byte codeword loads");
    static_assert(MC % U == 0, "a compile-time M is a whole number of batches");
    const int M = MC ? MC : Mrt;             // MC != 0: M known at compile time -- no bounds test between the loads, the batches unroll
#pragma unroll
    for (int m0 = 0; m0 < M; m0 += U) {
        float2 cv[U][DS / 2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float2 *src = reinterpret_cast<const float2 *>(codewords + ((size_t) (m0 + u < M ? m0 + u : m0) * 256 + tid) * DS);
#pragma unroll
            for (int i = 0; i < DS / 2; ++i) cv[u][i] = src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (m0 + u >= M) break;
            float x[DS], y[DS];
#pragma unroll
            for (int i = 0; i < DS; ++i) x[i] = q[(m0 + u) * DS + i];
#pragma unroll
            for (int i = 0; i < DS / 2; ++i) { y[2 * i] = cv[u][i].x; y[2 * i + 1] = cv[u][i].y; }
            lds[(m0 + u) * 256 + tid] = fvec_l2sqr_body(x, y, DS, ARCH);
        }
    }
}
// The SIMD variant is decided ONCE, outside the loops (round 6): with the run-time switch inside fvec_l2sqr_regs every entry sat behind
// its own branches, the query's sub-vector was fetched by a global load in front of each of them, and the M x Ds loads of a thread
// became M dependent L2 round trips -- 9.5 of the 11.6 us a block of the reference's harness shape (M = 64, Ds = 2) spent on its table.
// Up to 4 floats the three variants are the same operations (one chunk into zeroed lanes: fma(t, t, +0) is the rounded square), so one
// copy serves them; from 5 on the SSE build's second accumulation is mul-then-add, the AVX builds' a fused multiply-add.
template <int DS, int U, int MC = 0>
__device__ __forceinline__ void table_rows_regs(float *__restrict__ lds, const float *__restrict__ q, const float *__restrict__ codewords,
                                                 int M, int arch, int tid)
{
    if (DS <= 4 || arch == RII_SIMD_AVX512) table_rows_regs_arch<DS, U, RII_SIMD_AVX512, MC>(lds, q, codewords, M, tid);
    else if (arch == RII_SIMD_AVX) table_rows_regs_arch<DS, U, RII_SIMD_AVX, MC>(lds, q, codewords, M, tid);
    else table_rows_regs_arch<DS, U, RII_SIMD_SSE, MC>(lds, q, codewords, M, tid);
}


// L2SquaredDistance of src/pqkmeans.cpp:164-173 as auto-vectorised by GCC -Ofast ([objcode] in the oracle).
__device__ __forceinline__ float hsum_tree(float *t, int w)
{
    while (w > 4) {
        w >>= 1;
        for (int i = 0; i < w; ++i) t[i] = __fadd_rn(t[w + i], t[i]);
    }
    float a = __fadd_rn(t[2], t[0]), b = __fadd_rn(t[3], t[1]);
    return __fadd_rn(b, a);
}

__device__ inline float l2sq_pqk_dev(const float *__restrict__ a, const float *__restrict__ b, int n, int arch)
{
    const int W = (arch == RII_SIMD_AVX512) ? 16 : (arch == RII_SIMD_AVX ? 8 : 4);
    const bool fused = (arch != RII_SIMD_SSE);
    int i = 0;
    float acc = 0.f;
    float lanes[16];
    if (n >= W) {
        for (int l = 0; l < W; ++l) lanes[l] = 0.f;
        for (; i + W <= n; i += W)
            for (int l = 0; l < W; ++l) lanes[l] = sq_acc(lanes[l], __fsub_rn(a[i + l], b[i + l]), fused);
        acc = hsum_tree(lanes, W);
    }
    const int H = W / 2;
    if (H >= 4 && n - i >= H) {
        for (int l = 0; l < H; ++l) {
            float d = __fsub_rn(a[i + l], b[i + l]);
            lanes[l] = __fmul_rn(d, d);
        }
        acc = __fadd_rn(acc, hsum_tree(lanes, H));
        i += H;
    }
    for (; i < n; ++i) acc = sq_acc(acc, __fsub_rn(a[i], b[i]), fused);
    return acc;
}

__device__ __forceinline__ size_t lut_index(int64_t b, int i, int MK, int QT)
{
    return ((size_t) (b / QT) * MK + i) * QT + (size_t) (b % QT);
}


// wave-wide minimum of a packed 64-bit key (hi = orderable distance, lo = index) on the DPP path -- row_shr 1, 2, 4, 8 inside each
// row of 16 lanes, then row_bcast 15 / 31 across the rows: six VALU instructions per 32-bit half and no LDS round trip
// (__shfl_xor compiles to ds_bpermute_b32).  Returns the minimum in EVERY lane (read back from lane 63).
__device__ __forceinline__ uint32_t wave_min_u32_l63(uint32_t v)
{
#define RII_MIN_STEP(CTRL, ROWS)                                                                              \
    {                                                                                                         \
        const uint32_t o = (uint32_t) __builtin_amdgcn_update_dpp((int) 0xffffffffu, (int) v, CTRL, ROWS, 0xf, false); \
        v = o < v ? o : v;                                                                                    \
    }
    RII_MIN_STEP(0x111, 0xf) RII_MIN_STEP(0x112, 0xf) RII_MIN_STEP(0x114, 0xf) RII_MIN_STEP(0x118, 0xf)
    RII_MIN_STEP(0x142, 0xa) RII_MIN_STEP(0x143, 0xc)
#undef RII_MIN_STEP
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long key)
{
    const uint32_t hi = (uint32_t) (key >> 32);
    const uint32_t mh = wave_min_u32_l63(hi);
    const uint32_t ml = wave_min_u32_l63(hi == mh ? (uint32_t) (key & 0xffffffffu) : 0xffffffffu);
    return ((unsigned long long) mh << 32) | ml;
}

// Four independent minima at once (round 5: ivf_quad_kernel selects for four queries side by side): the same DPP ladder with the four
// chains written in lockstep, so every step's four instructions are independent and fill each other's DPP / VALU wait states --
// four dependent ladders one after the other cost a wave four times the latency of one.
__device__ __forceinline__ void wave_min_u32_l63_x4(uint32_t (&v)[4])
{
#define RII_MIN_STEP4(CTRL, ROWS)                                                                              \
    {                                                                                                          \
        uint32_t o[4];                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
            o[q] = (uint32_t) __builtin_amdgcn_update_dpp((int) 0xffffffffu, (int) v[q], CTRL, ROWS, 0xf, false); \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) v[q] = o[q] < v[q] ? o[q] : v[q];                        \
    }
    RII_MIN_STEP4(0x111, 0xf) RII_MIN_STEP4(0x112, 0xf) RII_MIN_STEP4(0x114, 0xf) RII_MIN_STEP4(0x118, 0xf)
    RII_MIN_STEP4(0x142, 0xa) RII_MIN_STEP4(0x143, 0xc)
#undef RII_MIN_STEP4
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (uint32_t) __builtin_amdgcn_readlane((int) v[q], 63);
}
__device__ __forceinline__ void wave_min_u64_x4(unsigned long long (&key)[4])
{
    uint32_t hi[4], mh[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mh[q] = hi[q] = (uint32_t) (key[q] >> 32);
    wave_min_u32_l63_x4(mh);
#pragma unroll
    for (int q = 0; q < 4; ++q) lo[q] = hi[q] == mh[q] ? (uint32_t) (key[q] & 0xffffffffu) : 0xffffffffu;
    wave_min_u32_l63_x4(lo);
#pragma unroll
    for (int q = 0; q < 4; ++q) key[q] = ((unsigned long long) mh[q] << 32) | lo[q];
}

// ---- block-local streaming top-k support (256 threads): LDS key buffer + bitonic sort ----
constexpr int kRrBuf = 2048;             // LDS key buffer; supports topk <= kRrBuf / 2

// sorts n keys (n a power of two) ascending in LDS with 256 threads
__device__ inline void rr_bitonic_sort(unsigned long long *buf, int tid, int n = kRrBuf)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < n / 2; t += 256) {
                const int i = 2 * t - (t & (stride - 1));           // lower index of the pair
                const int j = i + stride;
                const bool up = ((i & size) == 0);
                const unsigned long long x = buf[i], y = buf[j];
                if ((x > y) == up) { buf[i] = y; buf[j] = x; }
            }
        }
    }
    __syncthreads();
}


// ===================================================================================================
// std::partial_sort (libstdc++: __heap_select + __sort_heap over __adjust_heap/__push_heap), run by ONE
// lane per query on (id, dist) pairs compared on dist only -- the comparator of src/rii.h:234,279,312.
// Re-running the library's exact sequence of moves reproduces (i) which of several exactly tied
// candidates the reference returns and (ii) the order of the coarse lists *past* w, which QueryIvf walks
// when the first w lists hold fewer than topk hits (src/rii.h:283-326).
// ===================================================================================================
__device__ inline void pq_adjust_heap(int32_t *ids, float *ds, long hole, long len, int32_t vid, float vd)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (ds[child] < ds[child - 1]) child--;
        ids[hole] = ids[child]; ds[hole] = ds[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        ids[hole] = ids[child - 1]; ds[hole] = ds[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2;
    while (hole > top && ds[parent] < vd) {
        ids[hole] = ids[parent]; ds[hole] = ds[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    ids[hole] = vid; ds[hole] = vd;
}

__device__ inline void pq_partial_sort(int32_t *ids, float *ds, long middle, long n)
{
    long len = middle;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            pq_adjust_heap(ids, ds, parent, len, ids[parent], ds[parent]);
            if (parent == 0) break;
            parent--;
        }
    }
    if (len > 0) {
        for (long i = middle; i < n; ++i) {
            if (ds[i] < ds[0]) {
                const int32_t vid = ids[i];
                const float vd = ds[i];
                ids[i] = ids[0]; ds[i] = ds[0];
                pq_adjust_heap(ids, ds, 0, len, vid, vd);
            }
        }
    }
    while (len > 1) {
        --len;
        const int32_t vid = ids[len];
        const float vd = ds[len];
        ids[len] = ids[0]; ds[len] = ds[0];
        pq_adjust_heap(ids, ds, 0, len, vid, vd);
    }
}


// The same algorithm on PACKED entries (orderable distance bits << 32 | id) for working sets in LDS: one lane walks the
// heap, and every level of a sift costs one LDS round trip (the two children are adjacent: one ds_read2_b64) instead of
// two (distance, then id).  Comparisons look at the distance half only -- the reference's comparator (src/rii.h:234) --
// so the sequence of moves is libstdc++'s, move for move.
typedef unsigned long long pq64_t;
__device__ __forceinline__ pq64_t pq64_make(float d, uint32_t id) { return ((pq64_t) f32_orderable(__float_as_uint(d)) << 32) | id; }
__device__ __forceinline__ float pq64_dist(pq64_t e) { return __uint_as_float(f32_unorderable((uint32_t) (e >> 32))); }
__device__ __forceinline__ uint32_t pq64_id(pq64_t e) { return (uint32_t) (e & 0xffffffffu); }
__device__ __forceinline__ bool pq64_less(pq64_t a, pq64_t b) { return (uint32_t) (a >> 32) < (uint32_t) (b >> 32); }

__device__ __forceinline__ void pq64_adjust_heap(pq64_t *h, long hole, long len, pq64_t v)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        pq64_t a = h[child];
        const pq64_t b = h[child - 1];
        if (pq64_less(a, b)) { child--; a = b; }
        h[hole] = a;
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2;
    while (hole > top) {                                   // __push_heap
        const pq64_t pv = h[parent];
        if (!pq64_less(pv, v)) break;
        h[hole] = pv;
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h[hole] = v;
}

__device__ __forceinline__ void pq64_make_heap(pq64_t *h, long len)
{
    if (len < 2) return;
    long parent = (len - 2) / 2;
    for (;;) {
        pq64_adjust_heap(h, parent, len, h[parent]);
        if (parent == 0) break;
        parent--;
    }
}

__device__ __forceinline__ void pq64_sort_heap(pq64_t *h, long len)
{
    while (len > 1) {
        --len;
        const pq64_t v = h[len];
        h[len] = h[0];
        pq64_adjust_heap(h, 0, len, v);
    }
}

// std::partial_sort(first, first + middle, first + n) by ONE lane; the scan over [middle, n) fetches eight entries per LDS
// round trip (an entry that beats the heap top swaps with it in place, exactly like __pop_heap(first, middle, i))
__device__ __forceinline__ void pq64_partial_sort(pq64_t *h, long middle, long n)
{
    pq64_make_heap(h, middle);
    if (middle > 0) {
        pq64_t topv = h[0];
        for (long i0 = middle; i0 < n; i0 += 8) {
            pq64_t e[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) e[u] = (i0 + u < n) ? h[i0 + u] : ~0ull;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u < n && pq64_less(e[u], topv)) {
                    h[i0 + u] = topv;
                    pq64_adjust_heap(h, 0, middle, e[u]);
                    topv = h[0];
                }
            }
        }
    }
    pq64_sort_heap(h, middle);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same heap, walked by ONE WAVE (all 64 lanes active, every argument wave-uniform) instead of one lane.  A single lane
// pays a dependent chain of ~25 instructions plus an LDS round trip per LEVEL of a sift (~350 cycles; 7 levels for k = 100).
// libstdc++'s __adjust_heap first moves the hole down to a leaf along the larger children -- a path that depends only on
// the heap contents -- and then __push_heap lifts the value along that same path.  So:
//   1. every lane compares the two children of its nodes: one ballot per 64 nodes = the direction bit of every inner node;
//   2. the path p_0 .. p_L is followed on those bits (scalar code, no memory access);
//   3. lane j reads the old entry o_j of path node p_j (one LDS round trip for the whole path);
//   4. after the sift-down h[p_j] = o_{j+1}; __push_heap moves the hole back up while o_j < v (j = L, L-1, ..., 1): with
//      t = the level where that stops, the net effect is h[p_j] = o_{j+1} for j < t, h[p_t] = v, levels above t untouched.
// Same moves, same result, ~3 LDS round trips per adjust whatever the depth.  Heaps above 64 * kWhMaxWords inner nodes
// take the one-lane code.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kWhMaxWords = 8;           // direction words: heaps of up to 2 * 64 * 8 + 1 entries

// entry of the next lane (lanes 0..14 of a row; a path is at most 12 levels deep): DPP row_shl:1, no LDS round trip
__device__ __forceinline__ pq64_t wh_shfl_down1(pq64_t x)
{
    const uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (x & 0xffffffffu), 0x101, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (x >> 32), 0x101, 0xf, 0xf, false);
    return ((pq64_t) hi << 32) | lo;
}
__device__ __forceinline__ pq64_t wh_readlane(pq64_t x, int lane)
{
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (x & 0xffffffffu), lane);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (x >> 32), lane);
    return ((pq64_t) hi << 32) | lo;
}
__device__ __forceinline__ pq64_t wh_uniform(pq64_t x) { return wh_readlane(x, 0); }

// all 64 lanes of one wave; hole0, len, v uniform; NW = direction words (compile time: straight-line code, the NW pairs of
// LDS reads are in flight together): len <= 128 * NW + 1
template <int NW>
__device__ __forceinline__ pq64_t wh_adjust_heap(pq64_t *h, int hole0, int len, pq64_t v, int lane)
{
    // returns the entry at `hole0` afterwards (the new top when hole0 == 0): the caller needs no LDS read for it
    const int ninner = (len - 1) / 2;                       // nodes n < ninner have both children inside [0, len)
    unsigned long long W[NW];
    {
        pq64_t a[NW], b[NW];
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int n = lane + 64 * r;
            const int c = n < ninner ? 2 * n + 1 : 0;       // out-of-range lanes read entries 0 / 1: ignored below
            b[r] = h[c];
            a[r] = h[c + 1];
        }
#pragma unroll
        for (int r = 0; r < NW; ++r)                        // `if (comp(first + child, first + (child - 1))) child--`
            W[r] = __ballot(lane + 64 * r < ninner && pq64_less(a[r], b[r]));
    }
    // the path in scalar registers only: n_{j+1} = 2 n_j + 2 - bit_j, i.e. n_j + 1 = (hole0 + 1) 2^j + (the j-bit number of the
    // (1 - bit) choices so far); the lanes derive their own node from that number afterwards
    int n = hole0, L = 0;
    unsigned int choices = 0u;
    if constexpr (NW == 1) {
        const unsigned long long w = W[0];
        while (n < ninner) {                                // (the choices are read off the node number afterwards: half the scalar
            n = 2 * n + 2 - (int) ((w >> n) & 1ull);        //  instructions per level of the round-3 loop, which is a seventh of a sift)
            ++L;
        }
        choices = (unsigned int) (n + 1) - (((unsigned int) hole0 + 1u) << L);
    } else {                                                // word r lives in lane r: fetched with a scalar lane index
        int wlo = 0, whi = 0;
#pragma unroll
        for (int r = 0; r < NW; ++r)
            if (lane == r) { wlo = (int) (uint32_t) (W[r] & 0xffffffffu); whi = (int) (uint32_t) (W[r] >> 32); }
        while (n < ninner) {
            const int r = n >> 6, bitpos = n & 63;
            const uint32_t half = (uint32_t) (bitpos < 32 ? __builtin_amdgcn_readlane(wlo, r) : __builtin_amdgcn_readlane(whi, r));
            const unsigned int right = 1u - ((half >> (bitpos & 31)) & 1u);
            n = 2 * n + 1 + (int) right;
            choices = (choices << 1) | right;
            ++L;
        }
    }
    if ((len & 1) == 0 && n == (len - 2) / 2) {             // a last node with a single (left) child
        choices <<= 1;
        ++L;
    }
    const int mynode = lane <= L ? (int) ((((unsigned int) hole0 + 1u) << lane) + (choices >> (L - lane))) - 1 : 0;
    const pq64_t o = h[mynode];                             // lane 0's entry is the hole: never used
    const unsigned long long lessmask = __ballot(lane >= 1 && lane <= L && pq64_less(o, v));
    // t = L - (number of consecutive levels L, L-1, ... whose entry is < v), at least 0
    const unsigned long long range = ((2ull << L) - 1ull) & ~1ull;                            // bits 1 .. L (L <= 12)
    const unsigned long long stop = ~lessmask & range;      // levels that end the climb
    const int t = stop ? 63 - __builtin_clzll(stop) : 0;
    const pq64_t up = wh_shfl_down1(o);                     // o_{j+1}
    if (lane <= t) h[mynode] = lane < t ? up : v;
    __builtin_amdgcn_wave_barrier();                        // same wave, in-order LDS: later reads see these writes
    return t == 0 ? v : wh_readlane(o, 1);
}

template <int NW>
__device__ __forceinline__ void wh_partial_sort_t(pq64_t *h, int middle, int n, int lane)
{
    if (middle >= 2)                                        // __make_heap
        for (int parent = (middle - 2) / 2; parent >= 0; --parent) wh_adjust_heap<NW>(h, parent, middle, wh_uniform(h[parent]), lane);
    if (middle > 0) {
        pq64_t topv = wh_uniform(h[0]);
        for (int i0 = middle; i0 < n; i0 += 64) {           // __heap_select: 64 entries per LDS round trip
            const int i = i0 + lane;
            const pq64_t e = h[i < n ? i : 0];
            unsigned long long m = __ballot(i < n && pq64_less(e, topv));
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1ull;
                const pq64_t ej = wh_readlane(e, j);
                if (pq64_less(ej, topv)) {                  // the library compares with the top of THAT moment
                    if (lane == 0) h[i0 + j] = topv;        // __pop_heap(first, middle, i)
                    topv = wh_adjust_heap<NW>(h, 0, middle, ej, lane);
                }
            }
        }
    }
    pq64_t top = middle > 0 ? wh_uniform(h[0]) : 0ull;
    for (int len = middle - 1; len >= 1; --len) {           // __sort_heap
        const pq64_t v = wh_uniform(h[len]);
        if (lane == 0) h[len] = top;
        top = wh_adjust_heap<NW>(h, 0, len, v, lane);
    }
}

// std::partial_sort(first, first + middle, first + n) by one wave (all 64 lanes call it with uniform arguments)
__device__ __forceinline__ void wh_partial_sort(pq64_t *h, int middle, int n, int lane)
{
    if (middle <= 129) wh_partial_sort_t<1>(h, middle, n, lane);
    else if (middle <= 2 * 64 * kWhMaxWords) wh_partial_sort_t<kWhMaxWords>(h, middle, n, lane);
    else {                                                  // deeper heaps than the direction words cover: one-lane code
        if (lane == 0) pq64_partial_sort(h, middle, n);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// The same over a sequence that does not fit LDS: entries [0, middle) -- the heap -- live at `head` (LDS), entries [middle, n) at
// `tail[i - middle]` (global memory).  __heap_select only ever READS a tail entry once, at its own step, and writes the evicted heap
// top back into that very slot (__pop_heap(first, middle, i)); nothing reads a tail slot again before the call returns, so the tail
// needs no ordering beyond the caller's barrier afterwards.  middle <= 2 * 64 * kWhMaxWords + 1 (the caller checks).
template <int NW>
__device__ __forceinline__ void wh_partial_sort_split_t(pq64_t *head, pq64_t *tail, int middle, int n, int lane)
{
    if (middle >= 2)
        for (int parent = (middle - 2) / 2; parent >= 0; --parent) wh_adjust_heap<NW>(head, parent, middle, wh_uniform(head[parent]), lane);
    if (middle > 0) {
        pq64_t topv = wh_uniform(head[0]);
        // round 5: eight 64-entry slices of the tail are requested together (a tail slot is only ever written at its own step, after
        // it has been read: reading ahead sees what the step will see) -- one global round trip per 512 entries instead of per 64:
        // 11 k coarse lists used to cost a database-sharded query 175 dependent round trips before its first candidate
        constexpr int U = 8;
        for (int i0 = middle; i0 < n; i0 += 64 * U) {
            pq64_t ev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 64 * u + lane;
                ev[u] = i < n ? tail[i - middle] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 64 * u + lane;
                const pq64_t e = ev[u];
                unsigned long long m = __ballot(i < n && pq64_less(e, topv));
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const pq64_t ej = wh_readlane(e, j);
                    if (pq64_less(ej, topv)) {
                        if (lane == 0) tail[i0 + 64 * u + j - middle] = topv;
                        topv = wh_adjust_heap<NW>(head, 0, middle, ej, lane);
                    }
                }
            }
        }
    }
    pq64_t top = middle > 0 ? wh_uniform(head[0]) : 0ull;
    for (int len = middle - 1; len >= 1; --len) {
        const pq64_t v = wh_uniform(head[len]);
        if (lane == 0) head[len] = top;
        top = wh_adjust_heap<NW>(head, 0, len, v, lane);
    }
}
// (Round 5 measured a REGISTER-resident heap -- entry j in lane j, libstdc++'s __adjust_heap as wave-uniform scalar code over
//  v_readlane and compare-and-select writes -- against wh_adjust_top above: tools/ubench/heap_sift.hip, k = 100, 1100 sifts: 2297 cycles
//  per sift against 556 with the heap in LDS.  Every level of the scalar walk is a vector-to-scalar-to-vector round trip; the LDS
//  form resolves a whole path with one ballot and one gather.  Not kept.)

// (Round 5 also measured the WHOLE BLOCK scanning the tail of this partial_sort -- wave 0 replays the first 512 tail entries, publishes the
//  top it reached, 256 threads filter the rest of the tail against it and wave 0 replays the few survivors in sequence order: 28 us per
//  query over 8000 lists against 26-30 us for the split form above, and its 151 VGPRs cost ivf_shard_any_kernel its fourth block per
//  CU (39 -> 54 us per 1024 queries at SIFT shape).  Removed; the fast selection of ivfshard.hip makes the replay the rare route.)

constexpr int kWhSplitMaxHeap = 2 * 64 * kWhMaxWords;
__device__ __forceinline__ void wh_partial_sort_split(pq64_t *head, pq64_t *tail, int middle, int n, int lane)
{
    if (middle <= 129) wh_partial_sort_split_t<1>(head, tail, middle, n, lane);
    else wh_partial_sort_split_t<kWhMaxWords>(head, tail, middle, n, lane);
}

// single operations for callers that run the phases themselves (tieorder.hip); k <= 2 * 64 * kWhMaxWords
__device__ __forceinline__ pq64_t wh_adjust_top(pq64_t *h, int len, pq64_t v, int lane)        // returns the new top
{
    if (len <= 129) return wh_adjust_heap<1>(h, 0, len, v, lane);
    return wh_adjust_heap<kWhMaxWords>(h, 0, len, v, lane);
}
__device__ __forceinline__ void wh_make_heap(pq64_t *h, int len, int lane)
{
    if (len < 2) return;
    for (int parent = (len - 2) / 2; parent >= 0; --parent) {
        const pq64_t v = wh_uniform(h[parent]);
        if (len <= 129) wh_adjust_heap<1>(h, parent, len, v, lane);
        else wh_adjust_heap<kWhMaxWords>(h, parent, len, v, lane);
    }
}
__device__ __forceinline__ void wh_sort_heap(pq64_t *h, int len0, int lane)
{
    pq64_t top = len0 > 0 ? wh_uniform(h[0]) : 0ull;
    for (int len = len0 - 1; len >= 1; --len) {
        const pq64_t v = wh_uniform(h[len]);
        if (lane == 0) h[len] = top;
        top = wh_adjust_top(h, len, v, lane);
    }
}

// A bound T for the k1-th smallest of n 32-bit values held by the block (val(i), i < n: an LDS read), found with 256-bin histograms
// over the occupied range instead of a 32-step bisection (three block barriers per step): k1 <= #{val <= T}, and either
// #{val <= T} <= cap or T IS the k1-th smallest value.  One histogram pass when the bin of the k1-th smallest value ends at most
// `cap` values into the order (k1 << n: always); otherwise that bin is split into 256 again.  All threads of the block call it
// (barriers inside); s_hist = 256 words, s_ctl = 5 words of LDS.  Requires 1 <= k1 <= n.
template <typename Val>
__device__ __forceinline__ uint32_t block_kth_bound(Val val, int n, uint32_t k1, uint32_t cap, unsigned int *s_hist, unsigned int *s_ctl)
{
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
    if (tid == 0) { s_ctl[0] = 0xffffffffu; s_ctl[1] = 0u; }
    uint32_t vmin = 0xffffffffu, vmax = 0u;
    for (int i = tid; i < n; i += nt) {
        const uint32_t u = val(i);
        vmin = u < vmin ? u : vmin;
        vmax = u > vmax ? u : vmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t a = (uint32_t) __shfl_xor((int) vmin, off), c = (uint32_t) __shfl_xor((int) vmax, off);
        vmin = a < vmin ? a : vmin;
        vmax = c > vmax ? c : vmax;
    }
    __syncthreads();
    if (lane == 0) { atomicMin(&s_ctl[0], vmin); atomicMax(&s_ctl[1], vmax); }
    __syncthreads();
    uint32_t lo = s_ctl[0], below = 0u;               // `below` values lie under lo
    int shift;
    {
        const uint32_t span = s_ctl[1] - lo;
        const int bits = span ? 32 - __clz((int) span) : 0;
        shift = bits > 8 ? bits - 8 : 0;              // (val - lo) >> shift < 256 for every value
    }
    for (;;) {
        __syncthreads();
        if (tid < 256) s_hist[tid] = 0u;
        if (nt < 256) for (int i = tid + nt; i < 256; i += nt) s_hist[i] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            const uint32_t u = val(i);
            if (u >= lo && ((u - lo) >> shift) < 256u) atomicAdd(&s_hist[(u - lo) >> shift], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const uint32_t c0 = s_hist[4 * lane], c1 = s_hist[4 * lane + 1], c2 = s_hist[4 * lane + 2], c3 = s_hist[4 * lane + 3];
            const uint32_t sum = c0 + c1 + c2 + c3;
            uint32_t incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t) __shfl_up((int) incl, off);
                if (lane >= off) incl += t;
            }
            const uint32_t need = k1 - below, excl = incl - sum;
            if (excl < need && need <= incl) {        // exactly one lane: the pass holds at least `need` values
                uint32_t before = excl, cnt = c0;
                int bin = 4 * lane;
                if (before + cnt < need) { before += cnt; cnt = c1; ++bin; }
                if (before + cnt < need) { before += cnt; cnt = c2; ++bin; }
                if (before + cnt < need) { before += cnt; cnt = c3; ++bin; }
                s_ctl[2] = (uint32_t) bin; s_ctl[3] = before; s_ctl[4] = cnt;
            }
        }
        __syncthreads();
        const uint32_t bin = s_ctl[2], before = below + s_ctl[3], through = before + s_ctl[4];
        const uint32_t base = lo + (bin << shift);
        if (through <= cap || shift == 0) return base + ((1u << shift) - 1u);
        below = before;
        lo = base;
        shift = shift > 8 ? shift - 8 : 0;
    }
}

// sequential fp32 ADC (RiiCpp::ADist, src/rii.h:386-394) of one code against a plain [M][Ks] table in LDS
__device__ __forceinline__ float exact_adist(const float *lds, const uint8_t *code, int M, int Ks)
{
    float dist = 0.f;
    if ((M & 3) == 0) {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(code);
        for (int i = 0; i < M / 4; ++i) {
            const uint32_t w = cw[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) dist = __fadd_rn(dist, lds[(i * 4 + j) * Ks + ((w >> (8 * j)) & 0xffu)]);
        }
    } else {
        for (int m = 0; m < M; ++m) dist = __fadd_rn(dist, lds[m * Ks + code[m]]);
    }
    return dist;
}


}  // namespace riiamd
