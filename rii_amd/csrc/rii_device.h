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
__device__ inline float fvec_l2sqr_dev(const float *__restrict__ x, const float *__restrict__ y, int d, int arch)
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

// Ds == 4 (the SIFT shape: D=128, M=32) in straight-line form with two 16-byte loads.  Identical for all three SIMD
// variants: one 4-wide chunk into zeroed lanes (fma(t,t,+0) == the rounded square == SSE's mul-then-add-to-zero), no
// tail, then (l0+l1)+(l2+l3) (distance.h:148-169 / :194-216 / :229-251).  Callers guarantee 16-byte alignment.
__device__ __forceinline__ float fvec_l2sqr_ds4v(const float4 &a, const float4 &c)
{
    const float t0 = __fsub_rn(a.x, c.x), t1 = __fsub_rn(a.y, c.y), t2 = __fsub_rn(a.z, c.z), t3 = __fsub_rn(a.w, c.w);
    return __fadd_rn(__fadd_rn(__fmul_rn(t0, t0), __fmul_rn(t1, t1)), __fadd_rn(__fmul_rn(t2, t2), __fmul_rn(t3, t3)));
}
__device__ __forceinline__ float fvec_l2sqr_ds4(const float *__restrict__ x, const float *__restrict__ y)
{
    const float4 a = *reinterpret_cast<const float4 *>(x);
    const float4 c = *reinterpret_cast<const float4 *>(y);
    const float t0 = __fsub_rn(a.x, c.x), t1 = __fsub_rn(a.y, c.y), t2 = __fsub_rn(a.z, c.z), t3 = __fsub_rn(a.w, c.w);
    return __fadd_rn(__fadd_rn(__fmul_rn(t0, t0), __fmul_rn(t1, t1)), __fadd_rn(__fmul_rn(t2, t2), __fmul_rn(t3, t3)));
}

// dispatcher used by the table kernels (x = query sub-vector at m*Ds floats, y = codeword at i*Ds floats: both 16-byte
// aligned when Ds == 4 because the query rows and the codeword array are)
__device__ __forceinline__ float fvec_l2sqr_any(const float *__restrict__ x, const float *__restrict__ y, int d, int arch)
{
    if (d == 4) return fvec_l2sqr_ds4(x, y);
    return fvec_l2sqr_dev(x, y, d, arch);
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
