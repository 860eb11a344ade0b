// fastscan.hip -- top-1 linear ADC scan as an exact two-stage search (gfx950).
//
// The exact scan (kernels.hip: scan_kernel) is bound by LDS bank conflicts on 4-byte table entries: one
// ds_read_b128 serves 4 queries for one code byte.  Here the SAME ds_read_b128 serves 16 queries: stage 1
// scans all codes against an 8-bit quantisation of every query's table and keeps, per query, only the codes whose
// quantised sum is within a *proven* slack of the smallest quantised sum; stage 2 re-evaluates those few
// candidates with the exact fp32 table in the reference's order (RiiCpp::ADist, src/rii.h:386-394) and takes the
// (dist, id) minimum.  The result is bit-identical to scan_kernel's (and so to RiiCpp::QueryLinear for topk=1).
//
// Proof sketch (details next to lut_quantize_kernel): with T[m][ks] = lo_m + delta*c[m][ks] + r[m][ks],
//   d(n) = sum_m lo_m + delta*a(n) + sum_m r[m][code_m(n)],  a(n) = sum_m c[m][code_m(n)]  (integer).
// Rlo = sum_m min_ks r, Rhi = sum_m max_ks r bound the residual term for EVERY code; eps bounds the difference between
// the real-number sum d(n) and its sequentially rounded fp32 value.  If n* is the exact winner and n0 minimises a(),
// then a(n*) <= a(n0) + floor((Rhi - Rlo + 2*eps)/delta) =: a_min + slack.  All codes tied with n* at the exact minimum
// satisfy the same inequality, so the (dist, id) minimum over the candidate set equals the one over all codes.
#include "rii_internal.h"
#include "rii_device.h"
#include <float.h>
#include <algorithm>
#include <type_traits>

namespace riiamd {

constexpr int kFsThreads = 1024;
constexpr size_t kFsLdsBytes = 160 * 1024 - 512;    // LDS one scan block may use (tables + thresholds + staged candidates)
int fastscan_rows(int M, int Ks);      // queries per LDS row of the byte tables: 16, 8 or 0 (unsupported shape)
#ifndef RII_FS_LEVELS
#define RII_FS_LEVELS 63
#endif
constexpr int kFsLevels = RII_FS_LEVELS;            // quantisation levels - 1 (63: 6 bits, 31: 5 bits) ...
constexpr int kFsFlush = 255 / kFsLevels;            // ... so that kFsFlush entries add up inside a byte (4 x 63, 8 x 31)
static_assert(kFsFlush * kFsLevels <= 255 && (kFsFlush == 4 || kFsFlush == 8), "byte-packed partial sums must not carry");

// ---------------------------------------------------------------------------------------------------
// per-query quantisation of the exact table to kFsLevels+1 levels (one byte per entry) + slack.  One block (256 threads) per query.
//   qlut layout: [tile = b/16][m][ks][16 queries] u8
// ---------------------------------------------------------------------------------------------------
// shared body: T(i) returns the exact fp32 entry i = m*Ks + ks of query b's table
template <typename Getter>
__device__ __forceinline__ void quantize_table(const Getter &T, int64_t b, int M, int Ks,
                                               uint8_t *__restrict__ qc, int32_t *__restrict__ slack, int levels = kFsLevels)
{
    // levels (round 6): 63 = the byte tables every filter kernel understands; 127 / 255 for the matrix-core scans (fscan_mx_*), 255 stored
    // as level - 128 (signed bytes, the scan's accumulators start at 128 M: qlut_fused_kernel).  The slack below comes from the MEASURED
    // residuals of the levels actually stored, so it is proven for any step.  (Deep1B shape, Ds = 6: this generic path served it with
    // 63 levels -- a filter 4x coarser than the fused tables' -- and a structured 10 M-vector set sent thousands of codes per query
    // through the candidate path: 157 ms per 1024 queries, profiles/r06_deep_structured*.json.)
    __shared__ float s_lo[256], s_hi[256];          // per-m extrema (M <= 256)
    __shared__ double s_rlo[256], s_rhi[256];
    __shared__ float s_delta;
    const int tid = threadIdx.x;
    const int MK = M * Ks;
    const int wave = tid >> 6, lane = tid & 63;
    // 1. per-m min / max: one wave per m, lanes over ks
    for (int m = wave; m < M; m += 4) {
        float lo = INFINITY, hi = -INFINITY;
        for (int ks = lane; ks < Ks; ks += 64) {
            const float t = T(m * Ks + ks);
            lo = fminf(lo, t);
            hi = fmaxf(hi, t);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if (lane == 0) { s_lo[m] = lo; s_hi[m] = hi; }
    }
    __syncthreads();
    if (tid == 0) {
        float range = 0.f;
        for (int m = 0; m < M; ++m) range = fmaxf(range, s_hi[m] - s_lo[m]);
        float d = range / (float) levels;
        if (!(d > 0.f) || !isfinite(d)) d = 1.0f;
        s_delta = d * 1.000001f;
    }
    __syncthreads();
    const float delta = s_delta;
    const double ddelta = (double) delta;
    const int sub = levels > 127 ? 128 : 0;
    // 2. codes + residual extrema per m
    uint8_t *dst = qc + (size_t) b * MK;          // compact [b][M*Ks]: coalesced byte stores; qlut_interleave_kernel
                                                  // then builds the [tile][M*Ks][QR] rows the scan reads
    for (int m = wave; m < M; m += 4) {
        const float lo = s_lo[m];
        double rlo = INFINITY, rhi = -INFINITY;
        for (int ks = lane; ks < Ks; ks += 64) {
            const float t = T(m * Ks + ks);
            const float x = floorf((t - lo) / delta + 0.5f);
            const int c = (x >= (float) levels) ? levels : (x > 0.f ? (int) x : 0);
            dst[m * Ks + ks] = (uint8_t) (c - sub);
            const double r = (double) t - ((double) lo + (double) c * ddelta);
            rlo = fmin(rlo, r);
            rhi = fmax(rhi, r);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rlo = fmin(rlo, __shfl_xor(rlo, off));
            rhi = fmax(rhi, __shfl_xor(rhi, off));
        }
        if (lane == 0) { s_rlo[m] = rlo; s_rhi[m] = rhi; }
    }
    __syncthreads();
    if (tid == 0) {
        double Rlo = 0.0, Rhi = 0.0, dmax = 0.0;
        for (int m = 0; m < M; ++m) { Rlo += s_rlo[m]; Rhi += s_rhi[m]; dmax += fabs((double) s_hi[m]) + fabs((double) s_lo[m]); }
        // |fp32 sequential sum - real sum| <= (M-1) * 2^-24 * sum|t| (standard bound); use M * 2^-23 * dmax
        const double eps = (double) M * 1.1920928955078125e-07 * dmax;
        double s = (Rhi - Rlo + 2.0 * eps) / ddelta;
        s = s * (1.0 + 1e-9) + 2.0;                      // margins for the roundings of this very computation
        int32_t si = (s >= 0.0 && s < 60000.0) ? (int32_t) s : 60000;      // >= 0xffff - max a(): everything is a candidate
        slack[b] = si;
    }
}

struct GlobalLutGetter {
    const float *src; int QT;
    __device__ __forceinline__ float operator()(int i) const { return src[(size_t) i * QT]; }
};
struct LdsLutGetter {
    const float *lds;
    __device__ __forceinline__ float operator()(int i) const { return lds[i]; }
};

__global__ __launch_bounds__(256) void lut_quantize_kernel(const float *__restrict__ lut, int64_t B, int M, int Ks,
                                                           int QT, uint8_t *__restrict__ qc,
                                                           int32_t *__restrict__ slack, int levels)
{
    const int64_t b = blockIdx.x;
    GlobalLutGetter g{lut + (size_t) (b / QT) * M * Ks * QT + (b % QT), QT};
    quantize_table(g, b, M, Ks, qc, slack, levels);
}

// The common shapes (Ds == 4, M <= 32, Ks <= 256) keep the whole table of a query in registers: wave w owns the subspaces
// m = w, w+4, ... (8 of them), lane l the entries ks = l, l+64, ... (4 of them) -- 32 values per thread.  Building,
// per-subspace extrema, quantisation and residual extrema all run on those registers (no LDS copy of the table, no second
// and third pass over it), the eight per-subspace shuffle reductions of a wave are interleaved, and the division by the step
// is a multiplication by its reciprocal (any rounding of the level is fine: the residuals are taken from the levels actually
// stored).  One block used to take ~35 us whatever the batch size -- a fixed cost under every batch.
__device__ __forceinline__ void build_quantize_regs(const float *__restrict__ q, const float *__restrict__ codewords, int64_t b,
                                                    int M, int Ks, float *__restrict__ lut,
                                                    uint8_t *__restrict__ qc, int32_t *__restrict__ slack)
{
    __shared__ float s_lo[32], s_hi[32];
    __shared__ double s_rlo[32], s_rhi[32];
    __shared__ float s_delta;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int MK = M * Ks;
    float t[8][4];
    {
        const float4 *cw4 = reinterpret_cast<const float4 *>(codewords);
        const float4 *q4 = reinterpret_cast<const float4 *>(q);
#pragma unroll
        for (int h = 0; h < 2; ++h) {               // two halves: 16 codeword loads in flight, 64 VGPRs of staging
            float4 cv[4][4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = wave + 4 * (4 * h + mi), ks = lane + 64 * e;
                    cv[mi][e] = (m < M && ks < Ks) ? cw4[m * Ks + ks] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int m = wave + 4 * (4 * h + mi);
                const float4 qm = m < M ? q4[m] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[4 * h + mi][e] = fvec_l2sqr_ds4v(qm, cv[mi][e]);
            }
        }
    }
    float lo[8], hi[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = wave + 4 * mi;
        lo[mi] = INFINITY; hi[mi] = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ks = lane + 64 * e;
            if (m < M && ks < Ks) {
                lut[(size_t) b * MK + m * Ks + ks] = t[mi][e];        // plain [b][M*Ks] layout: coalesced, what the re-rank stages
                lo[mi] = fminf(lo[mi], t[mi][e]);
                hi[mi] = fmaxf(hi[mi], t[mi][e]);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            lo[mi] = fminf(lo[mi], __shfl_xor(lo[mi], off));
            hi[mi] = fmaxf(hi[mi], __shfl_xor(hi[mi], off));
        }
    if (lane == 0) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = wave + 4 * mi;
            if (m < M) { s_lo[m] = lo[mi]; s_hi[m] = hi[mi]; }
        }
    }
    __syncthreads();
    if (tid < 64) {
        float range = (tid < M) ? s_hi[tid] - s_lo[tid] : 0.f;
        // NaN ranges must poison the result like the serial fmaxf chain of quantize_table would not: keep its semantics
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) range = fmaxf(range, __shfl_xor(range, off));
        if (tid == 0) {
            float d = range / (float) kFsLevels;
            if (!(d > 0.f) || !isfinite(d)) d = 1.0f;
            s_delta = d * 1.000001f;
        }
    }
    __syncthreads();
    const float delta = s_delta, inv = 1.0f / delta;
    const double ddelta = (double) delta;
    uint8_t *dst = qc + (size_t) b * MK;          // compact [b][M*Ks]; qlut_interleave_kernel builds the rows the scan reads
    double rlo[8], rhi[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = wave + 4 * mi;
        rlo[mi] = INFINITY; rhi[mi] = -INFINITY;
        const float l = lo[mi];                    // every lane holds the reduced value after the xor-shuffles
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ks = lane + 64 * e;
            if (m < M && ks < Ks) {
                const float x = floorf((t[mi][e] - l) * inv + 0.5f);
                const int c = (x >= (float) kFsLevels) ? kFsLevels : (x > 0.f ? (int) x : 0);
                dst[m * Ks + ks] = (uint8_t) c;
                const double res = (double) t[mi][e] - ((double) l + (double) c * ddelta);
                rlo[mi] = fmin(rlo[mi], res);
                rhi[mi] = fmax(rhi[mi], res);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            rlo[mi] = fmin(rlo[mi], __shfl_xor(rlo[mi], off));
            rhi[mi] = fmax(rhi[mi], __shfl_xor(rhi[mi], off));
        }
    if (lane == 0) {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = wave + 4 * mi;
            if (m < M) { s_rlo[m] = rlo[mi]; s_rhi[m] = rhi[mi]; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        double Rlo = 0.0, Rhi = 0.0, dmax = 0.0;
        for (int m = 0; m < M; ++m) { Rlo += s_rlo[m]; Rhi += s_rhi[m]; dmax += fabs((double) s_hi[m]) + fabs((double) s_lo[m]); }
        // |fp32 sequential sum - real sum| <= (M-1) * 2^-24 * sum|t| (standard bound); use M * 2^-23 * dmax
        const double eps = (double) M * 1.1920928955078125e-07 * dmax;
        double sl = (Rhi - Rlo + 2.0 * eps) / ddelta;
        sl = sl * (1.0 + 1e-9) + 2.0;                      // margins for the roundings of this very computation
        slack[b] = (sl >= 0.0 && sl < 60000.0) ? (int32_t) sl : 60000;      // >= 0xffff - max a(): everything is a candidate
    }
}

// fused: exact table (fvec_L2sqr order, src/distance.h:117-252) -> global fp32 (for the re-rank) AND its quantisation
__global__ __launch_bounds__(256) void lut_build_quant_kernel(const float *__restrict__ queries, int64_t B,
                                                              const float *__restrict__ codewords, int M, int Ks, int Ds,
                                                              int arch, float *__restrict__ lut,
                                                              uint8_t *__restrict__ qc, int32_t *__restrict__ slack,
                                                              unsigned int *__restrict__ cand_cnt,
                                                              uint32_t *__restrict__ gthr, int levels)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_t = reinterpret_cast<float *>(smem);
    const int64_t b = blockIdx.x;
    const int MK = M * Ks;
    if (threadIdx.x == 0) {            // per-query state of the filter stage, reset here instead of by two memset launches
        if (cand_cnt) cand_cnt[b] = 0u;
        if (gthr) gthr[b] = 0xffffffffu;
    }
    const float *q = queries + b * (int64_t) (M * Ds);
    if (Ds == 4) {
        // 8 independent 16-byte codeword loads in flight per thread: a block per query has nothing else to hide the
        // load latency behind, and the one-entry-per-trip loop below was the ~30 us floor of small batches
        const float4 *cw4 = reinterpret_cast<const float4 *>(codewords);
        const float4 *q4 = reinterpret_cast<const float4 *>(q);
        for (int i0 = threadIdx.x; i0 < MK; i0 += 256 * 8) {
            float4 c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                c[u] = i < MK ? cw4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < MK) {
                    const float t = fvec_l2sqr_ds4v(q4[i / Ks], c[u]);
                    s_t[i] = t;
                    lut[(size_t) b * MK + i] = t;
                }
            }
        }
    } else
    for (int m = 0; m < M; ++m) {                    // query sub-vector address is wave-uniform inside this loop
        const float *qm = q + (size_t) m * Ds;
        const float *cm = codewords + (size_t) m * Ks * Ds;
        for (int ks = threadIdx.x; ks < Ks; ks += 256) {
            const int i = m * Ks + ks;
            const float t = fvec_l2sqr_any(qm, cm + (size_t) ks * Ds, Ds, arch);
            s_t[i] = t;
            lut[(size_t) b * MK + i] = t;          // plain [b][M*Ks] layout: coalesced, and what the re-rank stages
        }
    }
    __syncthreads();
    LdsLutGetter g{s_t};
    quantize_table(g, b, M, Ks, qc, slack, levels);
}

// the register-resident variant as its own kernel (its register budget must not be set by the generic paths above)
__global__ __launch_bounds__(256) void lut_build_quant_regs_kernel(const float *__restrict__ queries, int64_t B,
                                                                   const float *__restrict__ codewords, int M, int Ks,
                                                                   float *__restrict__ lut, uint8_t *__restrict__ qc,
                                                                   int32_t *__restrict__ slack,
                                                                   unsigned int *__restrict__ cand_cnt,
                                                                   uint32_t *__restrict__ gthr)
{
    const int64_t b = blockIdx.x;
    if (threadIdx.x == 0) {            // per-query state of the filter stage, reset here instead of by two memset launches
        if (cand_cnt) cand_cnt[b] = 0u;
        if (gthr) gthr[b] = 0xffffffffu;
    }
    build_quantize_regs(queries + b * (int64_t) (M * 4), codewords, b, M, Ks, lut, qc, slack);
}

// Rotated table layout (fs_rot_supported shapes).  The hardware serves a wave's ds_read_b128 in groups of G = 16 lanes
// (ds_read_b64: G = 32) and a row's bank slot is (row index mod G); with the plain [m][ks] order two lanes of a group
// collide whenever their code bytes share ks mod G -- 2.9 LDS cycles per group on random codes, 2.2 after the scan-order
// packing (scanorder.hip), against 1 conflict-free.  Here the rows of G consecutive subspaces are interleaved,
//     row(m, ks) = (m / G) * G * Ks + ks * G + (m mod G),
// so subspace m owns bank slot (m mod G), and lane l of a group works on subspace (l + t) mod G of ITS code at step t
// (the byte sums are integers: the order of the M additions is free).  The G lanes of a group then always hit G different
// slots: no bank conflict for any data.  The lane-dependent subspace order is baked into a formatted copy of the codes
// (fcodes_format_kernel): lookup t of half h of code n is the 16-bit value (h, ks, slot) whose shift by log2(row bytes)
// is the LDS address of the row.
__host__ __device__ __forceinline__ int fs_rot_row(int i, int Ks, int G)      // i = m * Ks + ks
{
    const int m = i / Ks, ks = i - m * Ks;
    return (m / G) * G * Ks + ks * G + (m % G);
}

// compact [b][M*Ks] bytes -> [tile][M*Ks][QR]: one 16-byte (QR=16) or 8-byte (QR=8) row per thread
template <int QR>
__global__ __launch_bounds__(256) void qlut_interleave_kernel(const uint8_t *__restrict__ qc, int64_t B, int MK, int Ks, int rot,
                                                              uint8_t *__restrict__ qlut)
{
    const int64_t tiles = (B + QR - 1) / QR;
    const int64_t total = tiles * MK;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t tile = t / MK;
        const int i = (int) (t - tile * MK);
        uint32_t w[QR / 4];
#pragma unroll
        for (int k = 0; k < QR / 4; ++k) w[k] = 0u;
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            const int64_t b = tile * QR + q;
            const uint32_t v = b < B ? qc[(size_t) b * MK + i] : 0u;
            w[q >> 2] |= v << (8 * (q & 3));
        }
        const int row = rot ? fs_rot_row(i, Ks, QR == 16 ? 16 : 32) : i;
        uint32_t *dst = reinterpret_cast<uint32_t *>(qlut + ((size_t) tile * MK + row) * QR);
#pragma unroll
        for (int k = 0; k < QR / 4; ++k) dst[k] = w[k];
    }
}

bool fs_rot_supported(int M, int Ks, int mx);

// rotated layout, 16-byte rows: one block per (tile, group of 16 subspaces, 16 consecutive ks).  Its 256 rows are
// contiguous in the destination (row = ks * 16 + q) and 16 runs of 16 bytes in every query's compact table (i = q * Ks + ks):
// the block reads those runs, turns the rows around in 4 KB of LDS and writes 4 KB coalesced (the generic kernel's stores
// were 256 bytes apart).
__global__ __launch_bounds__(256) void qlut_interleave_rot16_kernel(const uint8_t *__restrict__ qc, int64_t B, int MK,
                                                                    uint8_t *__restrict__ qlut)
{
    __shared__ uint4 s_rows[256];
    const int64_t tile = blockIdx.x;
    const int h = blockIdx.y, ks0 = blockIdx.z * 16;
    const int tid = threadIdx.x;
    const int q = tid >> 4, ks = ks0 + (tid & 15);
    const int i = h * 4096 + q * 256 + ks;                   // source index m * Ks + ks with m = 16 h + q
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t b = tile * 16 + j;
        const uint32_t v = b < B ? qc[(size_t) b * MK + i] : 0u;
        w[j >> 2] |= v << (8 * (j & 3));
    }
    s_rows[(tid & 15) * 16 + q] = make_uint4(w[0], w[1], w[2], w[3]);
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(qlut + ((size_t) tile * MK + (size_t) h * 4096 + (size_t) ks0 * 16) * 16);
    dst[tid] = s_rows[tid];
}

static hipError_t launch_qlut_interleave(const uint8_t *d_qc, int64_t B, int M, int Ks, uint8_t *d_qlut, int mx, hipStream_t st)
{
    const int qr = fastscan_rows(M, Ks);
    const int MK = M * Ks;
    const int rot = fs_rot_supported(M, Ks, mx) ? 1 : 0;
    if (rot && qr == 16 && Ks == 256) {
        hipLaunchKernelGGL(qlut_interleave_rot16_kernel, dim3((unsigned) ((B + 15) / 16), M / 16, 16), dim3(256), 0, st, d_qc, B, MK, d_qlut);
        return hipGetLastError();
    }
    const int64_t total = ((B + qr - 1) / qr) * MK;
    int blocks = (int) std::min<int64_t>((total + 255) / 256, 8192);
    if (qr == 16) hipLaunchKernelGGL(qlut_interleave_kernel<16>, dim3(blocks), dim3(256), 0, st, d_qc, B, MK, Ks, rot, d_qlut);
    else hipLaunchKernelGGL(qlut_interleave_kernel<8>, dim3(blocks), dim3(256), 0, st, d_qc, B, MK, Ks, rot, d_qlut);
    return hipGetLastError();
}

// codes [n][M] u8 -> formatted lookups [n][M] u16 for the rotated layout; ids == NULL: code n itself, else the code ids[n]
// (subset search).  Thread per (position, lookup).
__global__ __launch_bounds__(256) void fcodes_format_kernel(const uint8_t *__restrict__ codes, const int64_t *__restrict__ ids,
                                                            int64_t n0, int64_t n1, int M, int G, int sh,
                                                            uint16_t *__restrict__ out)
{
    const int64_t total = (n1 - n0) * M;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t n = n0 + t / M;
        const int i = (int) (t % M);                    // lookup index = half * G + step
        const int h = i / G, step = i - h * G;
        const int slot = ((int) (n & (G - 1)) + step) & (G - 1);
        const int64_t src = ids ? ids[n] : n;
        const uint32_t ks = codes[(size_t) src * M + h * G + slot];
        out[(size_t) n * M + i] = (uint16_t) (((uint32_t) h << (sh + 8)) | (ks << sh) | (uint32_t) slot);
    }
}

// fscan_mx_kernel's order of the lookups (defined next to that kernel)
__global__ void fcodes_mx_format_kernel(const uint8_t *__restrict__ codes, const int64_t *__restrict__ ids, int64_t n0, int64_t n1,
                                        int M, uint8_t *__restrict__ out);

// formats codes [n0, n1) (or the gathered codes ids[n0..n1)) for the filter scan of an (M, Ks) index
template <int M_>
__global__ void fcodes_mx_format_rows_kernel(const uint8_t *__restrict__ codes, const int64_t *__restrict__ ids, int64_t n0, int64_t n1,
                                             uint8_t *__restrict__ out);
hipError_t launch_fcodes_format(const uint8_t *d_codes, const int64_t *d_ids, int64_t n0, int64_t n1, int M, int Ks,
                                uint16_t *d_out, int mx, hipStream_t st)
{
    if (n1 <= n0) return hipSuccess;
    if (mx) {              // fscan_mx_kernel's order: whole groups of 16 codes, starting at the group n0 falls into
        n0 = n0 / 16 * 16;
        const int64_t tot = ((n1 + 15) / 16 * 16 - n0) * M / 4;         // one thread per output dword
        const int nb = (int) std::min<int64_t>((tot + 255) / 256, 16384);
        if (M == 16 || M == 32) {                                       // one thread per (group, lane): 16-byte row loads
            const int64_t slots = ((n1 + 15) / 16 * 16 - n0) * 4;
            const int nbr = (int) std::min<int64_t>((slots + 255) / 256, 16384);
            if (M == 32) hipLaunchKernelGGL(fcodes_mx_format_rows_kernel<32>, dim3(nbr), dim3(256), 0, st, d_codes, d_ids, n0, n1, reinterpret_cast<uint8_t *>(d_out));
            else hipLaunchKernelGGL(fcodes_mx_format_rows_kernel<16>, dim3(nbr), dim3(256), 0, st, d_codes, d_ids, n0, n1, reinterpret_cast<uint8_t *>(d_out));
            return hipGetLastError();
        }
        hipLaunchKernelGGL(fcodes_mx_format_kernel, dim3(nb), dim3(256), 0, st, d_codes, d_ids, n0, n1, M,
                           reinterpret_cast<uint8_t *>(d_out));
        return hipGetLastError();
    }
    const int G = fastscan_rows(M, Ks) == 16 ? 16 : 32;
    const int sh = G == 16 ? 4 : 5;                    // value = (half, ks, slot): ks above the log2(G) slot bits
    const int64_t total = (n1 - n0) * M;
    const int blocks = (int) std::min<int64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(fcodes_format_kernel, dim3(blocks), dim3(256), 0, st, d_codes, d_ids, n0, n1, M, G, sh, d_out);
    return hipGetLastError();
}

hipError_t launch_lut_quantize(const float *d_lut, int64_t B, int M, int Ks, int QT, uint8_t *d_qc, uint8_t *d_qlut,
                               int32_t *d_slack, int mx, hipStream_t st, int levels)
{
    if (B == 0) return hipSuccess;
    if (levels != 63 && levels != 127 && levels != 255) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lut_quantize_kernel, dim3((unsigned) B), dim3(256), 0, st, d_lut, B, M, Ks, QT, d_qc, d_slack, levels);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_qlut_interleave(d_qc, B, M, Ks, d_qlut, mx, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Tables of the rotated shapes (Ds = 4, Ks = 256, M = 16 / 32), built by TILE instead of by query.
//
// lut_build_quant_regs_kernel (block per query) reads the whole codebook (128 KiB) in every block: 134 MB of L2 traffic per
// 1024-query batch, a compact byte table per query and a second kernel to interleave 16 of them.  Here
//   1. lut_tile_build_kernel, grid (tile of 16 queries, group of 8 subspaces, quarter of the tile): a wave keeps the codewords
//      of its two subspaces in registers (8 float4) and runs four queries past them: exact fp32 entries (fvec_L2sqr order)
//      to the plain [b][M*Ks] table the re-rank stages, plus min / max per (query, subspace).  Codebook traffic: 32 MB.
//   2. qlut_tile_quant_kernel, grid (tile, 16 subspaces, 16 ks): quantisation step per query from the 16 x M extrema, the 6-bit
//      levels of 16 queries packed into one 16-byte row, rows turned around in 4 KB of LDS and written in the rotated order.
// The slack of a query no longer needs the measured residuals: with c = floor((t - lo_m) / delta + 1/2) evaluated in fp32 the
// level is off the real quotient y by at most 1/2 + 1.4e-5 (three roundings of values below 64), and (t - lo_m) / delta < 63
// keeps it inside [0, 63] without clamping, so every residual lies in delta * [-(1/2 + 1.4e-5), +(1/2 + 1.4e-5)] and
// Rhi - Rlo <= M * delta * (1 + 2.8e-5): slack = floor(M (1 + 1e-4) + 2 eps / delta) + 2 (the measured ranges were
// 0.99 delta per subspace anyway).
// ---------------------------------------------------------------------------------------------------------------------
template <typename Vec>          // float4: Ds = 4, float2: Ds = 2
__global__ __launch_bounds__(256) void lut_tile_build_kernel(const float *__restrict__ queries, int64_t B,
                                                             const float *__restrict__ codewords, int M,
                                                             float *__restrict__ lut, float2 *__restrict__ lohi)
{
    const int64_t tile = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int m0 = blockIdx.y * 8 + wave * 2;                // this wave: subspaces m0, m0 + 1
    const int MK = M * 256;
    constexpr int Ds = (int) (sizeof(Vec) / sizeof(float));
    const Vec *cw4 = reinterpret_cast<const Vec *>(codewords);
    Vec cv[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) cv[s][e] = cw4[(m0 + s) * 256 + lane + 64 * e];
    {                                                        // four queries per block: their 16 reductions travel together
        const int j0 = blockIdx.z * 4;
        float lo[4][2], hi[4][2];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int64_t b = tile * 16 + j0 + jj;
            const bool live = b < B;
            const Vec *q4 = reinterpret_cast<const Vec *>(queries + (live ? b : 0) * (int64_t) (M * Ds));
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Vec qm = q4[m0 + s];
                lo[jj][s] = INFINITY; hi[jj][s] = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = fvec_l2sqr_vec(qm, cv[s][e]);
                    if (live) lut[(size_t) b * MK + (m0 + s) * 256 + lane + 64 * e] = t;
                    lo[jj][s] = fminf(lo[jj][s], t);
                    hi[jj][s] = fmaxf(hi[jj][s], t);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    lo[jj][s] = fminf(lo[jj][s], __shfl_xor(lo[jj][s], off));
                    hi[jj][s] = fmaxf(hi[jj][s], __shfl_xor(hi[jj][s], off));
                }
        if (lane < 8) {                                      // lane = jj * 2 + s writes its pair
            const int jj = lane >> 1, s = lane & 1;
            const int64_t b = tile * 16 + j0 + jj;
            float l = lo[0][0], h = hi[0][0];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (a == jj && c == s) { l = lo[a][c]; h = hi[a][c]; }
            if (b < B) lohi[(size_t) b * M + m0 + s] = make_float2(l, h);
        }
    }
}

__global__ __launch_bounds__(256) void qlut_tile_quant_kernel(const float *__restrict__ lut, const float2 *__restrict__ lohi,
                                                              int64_t B, int M, uint8_t *__restrict__ qlut,
                                                              int32_t *__restrict__ slack, unsigned int *__restrict__ cand_cnt,
                                                              uint32_t *__restrict__ gthr)
{
    __shared__ uint4 s_rows[256];
    __shared__ float s_inv[16], s_delta[16];
    const int64_t tile = blockIdx.x;
    const int h = blockIdx.y, ks0 = blockIdx.z * 16;
    const int tid = threadIdx.x;
    const int MK = M * 256;
    {   // per-query quantisation step: 16 lanes per query reduce the ranges (and |lo| + |hi| for eps) over the M subspaces
        const int j = tid >> 4, l16 = tid & 15;
        const int64_t b = tile * 16 + j;
        float range = 0.f;
        double dmax = 0.0;
        if (b < B)
            for (int m = l16; m < M; m += 16) {
                const float2 lh = lohi[(size_t) b * M + m];
                range = fmaxf(range, lh.y - lh.x);
                dmax += fabs((double) lh.x) + fabs((double) lh.y);
            }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            range = fmaxf(range, __shfl_xor(range, off));
            dmax += __shfl_xor(dmax, off);
        }
        if (l16 == 0) {
            float d = range / (float) kFsLevels;
            if (!(d > 0.f) || !isfinite(d)) d = 1.0f;
            const float delta = d * 1.000001f;
            s_delta[j] = delta;
            s_inv[j] = 1.0f / delta;
            if (h == 0 && blockIdx.z == 0 && b < B) {        // one block per tile publishes the per-query state of the filter
                const double eps = (double) M * 1.1920928955078125e-07 * dmax;
                double sl = (double) M * (1.0 + 1e-4) + 2.0 * eps / (double) delta;
                sl = sl * (1.0 + 1e-9) + 2.0;
                slack[b] = (sl >= 0.0 && sl < 60000.0) ? (int32_t) sl : 60000;
                if (cand_cnt) cand_cnt[b] = 0u;
                if (gthr) gthr[b] = 0xffffffffu;
            }
        }
    }
    __syncthreads();
    const int q = tid >> 4, ks = ks0 + (tid & 15);
    const int m = h * 16 + q;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t b = tile * 16 + j;
        uint32_t c = 0u;
        if (b < B) {
            const float t = lut[(size_t) b * MK + m * 256 + ks];
            const float l = lohi[(size_t) b * M + m].x;
            const float x = floorf((t - l) * s_inv[j] + 0.5f);
            c = (x >= (float) kFsLevels) ? (uint32_t) kFsLevels : (x > 0.f ? (uint32_t) x : 0u);
        }
        w[j >> 2] |= c << (8 * (j & 3));
    }
    s_rows[(tid & 15) * 16 + q] = make_uint4(w[0], w[1], w[2], w[3]);
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(qlut + ((size_t) tile * MK + (size_t) h * 4096 + (size_t) ks0 * 16) * 16);
    dst[tid] = s_rows[tid];
}

// the same for M = 64 (8 queries per tile, 8-byte rows, 32 subspaces per rotated half: row = half * 8192 + ks * 32 + slot):
// grid (tile of 8 queries, half, 8 consecutive ks); the block's 256 rows are contiguous in the destination
__global__ __launch_bounds__(256) void qlut_tile_quant8_kernel(const float *__restrict__ lut, const float2 *__restrict__ lohi,
                                                               int64_t B, int M, uint8_t *__restrict__ qlut,
                                                               int32_t *__restrict__ slack, unsigned int *__restrict__ cand_cnt,
                                                               uint32_t *__restrict__ gthr)
{
    __shared__ uint2 s_rows[256];
    __shared__ float s_inv[8];
    const int64_t tile = blockIdx.x;
    const int h = blockIdx.y, ks0 = blockIdx.z * 8;
    const int tid = threadIdx.x;
    const int MK = M * 256;
    {   // per-query quantisation step: 32 lanes per query reduce the ranges (and |lo| + |hi| for eps) over the M subspaces
        const int j = tid >> 5, l32 = tid & 31;
        const int64_t b = tile * 8 + j;
        float range = 0.f;
        double dmax = 0.0;
        if (b < B)
            for (int m = l32; m < M; m += 32) {
                const float2 lh = lohi[(size_t) b * M + m];
                range = fmaxf(range, lh.y - lh.x);
                dmax += fabs((double) lh.x) + fabs((double) lh.y);
            }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            range = fmaxf(range, __shfl_xor(range, off));
            dmax += __shfl_xor(dmax, off);
        }
        if (l32 == 0) {
            float d = range / (float) kFsLevels;
            if (!(d > 0.f) || !isfinite(d)) d = 1.0f;
            const float delta = d * 1.000001f;
            s_inv[j] = 1.0f / delta;
            if (h == 0 && blockIdx.z == 0 && b < B) {        // one block per tile publishes the per-query state of the filter
                const double eps = (double) M * 1.1920928955078125e-07 * dmax;
                double sl = (double) M * (1.0 + 1e-4) + 2.0 * eps / (double) delta;
                sl = sl * (1.0 + 1e-9) + 2.0;
                slack[b] = (sl >= 0.0 && sl < 60000.0) ? (int32_t) sl : 60000;
                if (cand_cnt) cand_cnt[b] = 0u;
                if (gthr) gthr[b] = 0xffffffffu;
            }
        }
    }
    __syncthreads();
    const int slot = tid >> 3, ks = ks0 + (tid & 7);
    const int m = h * 32 + slot;
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t b = tile * 8 + j;
        uint32_t c = 0u;
        if (b < B) {
            const float t = lut[(size_t) b * MK + m * 256 + ks];
            const float l = lohi[(size_t) b * M + m].x;
            const float x = floorf((t - l) * s_inv[j] + 0.5f);
            c = (x >= (float) kFsLevels) ? (uint32_t) kFsLevels : (x > 0.f ? (uint32_t) x : 0u);
        }
        w[j >> 2] |= c << (8 * (j & 3));
    }
    s_rows[(tid & 7) * 32 + slot] = make_uint2(w[0], w[1]);
    __syncthreads();
    uint2 *dst = reinterpret_cast<uint2 *>(qlut + ((size_t) tile * MK + (size_t) h * 8192 + (size_t) ks0 * 32) * 8);
    dst[tid] = s_rows[tid];
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE launch for the tables of the matrix-core filter (round 3; M = 16 / 32, Ks = 256, Ds = 4 / 2): grid (tile of 16 queries,
// quarter of the tile), 1024 threads.  The 16 waves of a block cover ALL M subspaces (MW per wave) for the quarter's four queries,
// so the per-query quantisation step -- a maximum over every subspace -- is a block-local reduction and nothing has to travel
// through global memory between "build" and "quantise":
//   codewords of the wave's subspaces in registers (as lut_tile_build_kernel) -> the 32 exact entries of a thread stay in
//   registers -> min / max per (query, subspace) by wave shuffles -> LDS -> step, reciprocal, slack per query -> levels
//   c = floor((t - lo) / delta + 1/2) (the same fp32 expression as qlut_tile_quant_kernel: identical bytes) -> the four queries'
//   levels of an entry packed into ONE dword, stored coalesced as a "quarter table" [tile][quarter][m][ks] u32.
// fscan_mx_kernel scatters the four quarter tables of its tile into the 16-byte rotated rows while staging them (a wave writes 256
// contiguous LDS bytes per ds_write_b32: conflict-free).  The exact fp32 table goes to global memory only when somebody will read
// it (top-k re-rank, tie order); the top-1 re-rank rebuilds what it needs from the codebook (rerank_top1_direct_kernel).
// Cost at the bench shape: 256 blocks, 128 KiB of codebook (L2) in and 32 KiB out per block, no 32 MB fp32 round trip.
// ---------------------------------------------------------------------------------------------------------------------
// LEVELS = 63: the byte tables every filter kernel understands.  LEVELS = 255 (fscan_mx_* only): the matrix core adds SIGNED bytes
// without carries, so a table entry may use the whole byte -- stored as level - 128, the scan starts its accumulator at 128 M
// instead of 0 -- and the quantisation step is 4x finer: ~4x fewer codes fall within the proven slack of the running minimum
// (fewer trips through the scan's divergent candidate path, fewer candidates for the re-rank).  Rounding: the level is off the real
// quotient by at most 1/2 + 6.1e-5 (three fp32 roundings of values below 256), so Rhi - Rlo <= M delta (1 + 1.3e-4).
// wave-wide min / max on the DPP path (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast 15 / 31 across the rows): six
// VALU instructions per value and no LDS round trip; __shfl_xor compiles to ds_bpermute_b32 -- 96 dependent LDS operations for the 16
// extrema of a thread here.  The result is valid in LANE 63.
template <bool MAX> __device__ __forceinline__ float fs_wave_extremum_l63(float v)
{
    const int ident = MAX ? (int) 0xff800000u : 0x7f800000;             // -inf / +inf: lanes without a source keep their value
#define RII_STEP(CTRL, ROWS)                                                                                             \
    {                                                                                                                    \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(ident, __float_as_int(v), CTRL, ROWS, 0xf, false));   \
        v = MAX ? fmaxf(v, o) : fminf(v, o);                                                                             \
    }
    RII_STEP(0x111, 0xf) RII_STEP(0x112, 0xf) RII_STEP(0x114, 0xf) RII_STEP(0x118, 0xf)       // row_shr:1, 2, 4, 8
    RII_STEP(0x142, 0xa) RII_STEP(0x143, 0xc)                                                  // row_bcast:15 (rows 1, 3), row_bcast:31 (rows 2, 3)
#undef RII_STEP
    return v;
}

// NQ = 4: grid (tile, quarter), a block builds the quarter's four queries and stores whole dwords.  NQ = 1 (round 4, small batches):
// grid (tile, query of the tile) -- a quarter of the instructions per wave (the kernel is issue-bound inside a CU: 1011 vector
// instructions x 16 waves), four times the blocks: at B = 128 the tables take 128 CUs instead of 32.  The block owns ONE byte of
// every dword of its quarter table (byte stores; the other three bytes come from the blocks of the quarter's other queries).
// Identical bytes either way.
template <typename Vec, int MW, int LEVELS, int NQ>  // float4: Ds = 4, float2: Ds = 2; MW = subspaces per wave (M = 16 MW)
__global__ __launch_bounds__(1024) void qlut_fused_kernel(const float *__restrict__ queries, int64_t B,
                                                          const float *__restrict__ codewords, float *__restrict__ lut,
                                                          uint32_t *__restrict__ qlut4, int32_t *__restrict__ slack,
                                                          unsigned int *__restrict__ cand_cnt, uint32_t *__restrict__ gthr)
{
    constexpr int M = 16 * MW;
    constexpr int Ds = (int) (sizeof(Vec) / sizeof(float));
    constexpr int MK = M * 256;
    __shared__ float s_lo[NQ][M], s_hi[NQ][M];
    __shared__ float s_inv[NQ];
    const int64_t tile = blockIdx.x;
    const int quarter = NQ == 4 ? (int) blockIdx.y : (int) (blockIdx.y >> 2);
    const int j0 = NQ == 4 ? 0 : (int) (blockIdx.y & 3);          // first query of the quarter this block builds
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int m0 = wave * MW;
    const Vec *cw4 = reinterpret_cast<const Vec *>(codewords);
    Vec cv[MW][4];
#pragma unroll
    for (int s_ = 0; s_ < MW; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e) cv[s_][e] = cw4[(m0 + s_) * 256 + lane + 64 * e];
    float t[NQ][MW][4], lo[NQ][MW], hi[NQ][MW];
#pragma unroll
    for (int jj = 0; jj < NQ; ++jj) {
        const int64_t b = tile * 16 + quarter * 4 + j0 + jj;
        const Vec *q4 = reinterpret_cast<const Vec *>(queries + (b < B ? b : 0) * (int64_t) (M * Ds));
#pragma unroll
        for (int s_ = 0; s_ < MW; ++s_) {
            const Vec qm = q4[m0 + s_];
            lo[jj][s_] = INFINITY; hi[jj][s_] = -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fvec_l2sqr_vec(qm, cv[s_][e]);
                t[jj][s_][e] = v;
                lo[jj][s_] = fminf(lo[jj][s_], v);
                hi[jj][s_] = fmaxf(hi[jj][s_], v);
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < NQ; ++jj)
#pragma unroll
        for (int s_ = 0; s_ < MW; ++s_) {
            const float l = fs_wave_extremum_l63<false>(lo[jj][s_]), h = fs_wave_extremum_l63<true>(hi[jj][s_]);
            if (lane == 63) { s_lo[jj][m0 + s_] = l; s_hi[jj][m0 + s_] = h; }
        }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < NQ; ++jj)                      // every lane needs its subspaces' minima for the levels
#pragma unroll
        for (int s_ = 0; s_ < MW; ++s_) lo[jj][s_] = s_lo[jj][m0 + s_];
    double my_dmax = 0.0;
    float my_delta = 0.f;
    if (threadIdx.x < NQ * 32) {         // 32 lanes per query: the range over the M subspaces -> step and reciprocal
        const int j = threadIdx.x >> 5, l32 = threadIdx.x & 31;
        float range = 0.f;
        for (int m = l32; m < M; m += 32) {
            range = fmaxf(range, s_hi[j][m] - s_lo[j][m]);
            my_dmax += fabs((double) s_lo[j][m]) + fabs((double) s_hi[j][m]);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) range = fmaxf(range, __shfl_xor(range, off));
        float d = range / (float) LEVELS;
        if (!(d > 0.f) || !isfinite(d)) d = 1.0f;
        my_delta = d * 1.000001f;
        if (l32 == 0) s_inv[j] = 1.0f / my_delta;
    }
    __syncthreads();
    if (threadIdx.x < NQ * 32) {         // the per-query state of the filter stage, OFF the critical path of the other waves (double math)
        const int j = threadIdx.x >> 5, l32 = threadIdx.x & 31;
        const int64_t b = tile * 16 + quarter * 4 + j0 + j;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) my_dmax += __shfl_xor(my_dmax, off);
        if (l32 == 0 && b < B) {         // (see qlut_tile_quant_kernel for the slack)
            const double eps = (double) M * 1.1920928955078125e-07 * my_dmax;
            double sl = (double) M * (1.0 + (LEVELS > 63 ? 2e-4 : 1e-4)) + 2.0 * eps / (double) my_delta;
            sl = sl * (1.0 + 1e-9) + 2.0;
            slack[b] = (sl >= 0.0 && sl < 60000.0) ? (int32_t) sl : 60000;
            if (cand_cnt) cand_cnt[b] = 0u;
            if (gthr) gthr[b] = 0xffffffffu;
        }
    }
    uint32_t *dst = qlut4 + ((size_t) tile * 4 + quarter) * MK;
#pragma unroll
    for (int s_ = 0; s_ < MW; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t w = 0u;
#pragma unroll
            for (int jj = 0; jj < NQ; ++jj) {
                const int64_t b = tile * 16 + quarter * 4 + j0 + jj;
                uint32_t c = 0u;
                if (b < B) {
                    // x >= 1/2 always (t >= lo), so floor = the truncating conversion (which saturates and sends NaN to 0), then one
                    // clamp: the same level as floorf + a two-sided clamp in half the instructions (this kernel is VALU-bound:
                    // 1011 instructions per wave, four waves per SIMD)
                    const float x = (t[jj][s_][e] - lo[jj][s_]) * s_inv[jj] + 0.5f;
                    c = min(__float2uint_rz(x), (uint32_t) LEVELS);
                }
                if (LEVELS > 127) c ^= 0x80u;            // stored as the signed byte (level - 128); dead queries: -128, never judged
                w |= c << (8 * jj);
            }
            if constexpr (NQ == 4) dst[(m0 + s_) * 256 + lane + 64 * e] = w;
            else reinterpret_cast<uint8_t *>(dst)[(size_t) ((m0 + s_) * 256 + lane + 64 * e) * 4 + j0] = (uint8_t) w;
        }
    if (lut) {                            // the exact table, plain [b][M*Ks]: only for the callers that read it (top-k, tie order)
#pragma unroll
        for (int jj = 0; jj < NQ; ++jj) {
            const int64_t b = tile * 16 + quarter * 4 + j0 + jj;
            if (b < B)
#pragma unroll
                for (int s_ = 0; s_ < MW; ++s_)
#pragma unroll
                    for (int e = 0; e < 4; ++e) lut[(size_t) b * MK + (m0 + s_) * 256 + lane + 64 * e] = t[jj][s_][e];
        }
    }
}

bool qlut_fused_supported(int M, int Ks, int Ds, int mx) { return mx && Ks == 256 && (M == 16 || M == 32) && (Ds == 4 || Ds == 2) && fs_rot_supported(M, Ks, mx); }
size_t qlut_fused_bytes(int64_t B, int M) { return (size_t) ((B + 15) / 16) * 4 * M * 256 * 4; }
hipError_t launch_qlut_fused(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ds, float *d_lut_or_null,
                             uint32_t *d_qlut4, int32_t *d_slack, unsigned int *d_cand_cnt, uint32_t *d_gthr, int levels, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    // small batches: one query per block (four times the blocks, a quarter of the instructions per wave)
    const bool one = B <= 256;
    const dim3 grid((unsigned) ((B + 15) / 16), one ? 16 : 4), block(1024);
#define RII_QF(VEC, MW, LV) do { if (one) hipLaunchKernelGGL((qlut_fused_kernel<VEC, MW, LV, 1>), grid, block, 0, st, d_queries, B, d_codewords, d_lut_or_null, d_qlut4, d_slack, d_cand_cnt, d_gthr); \
                                 else hipLaunchKernelGGL((qlut_fused_kernel<VEC, MW, LV, 4>), grid, block, 0, st, d_queries, B, d_codewords, d_lut_or_null, d_qlut4, d_slack, d_cand_cnt, d_gthr); } while (0)
    if (levels != 63 && levels != 127 && levels != 255) return hipErrorInvalidValue;
#define RII_QF3(VEC, MW) { if (levels == 255) RII_QF(VEC, MW, 255); else if (levels == 127) RII_QF(VEC, MW, 127); else RII_QF(VEC, MW, 63); }
    if (M == 32 && Ds == 4) RII_QF3(float4, 2)
    else if (M == 16 && Ds == 4) RII_QF3(float4, 1)
    else if (M == 32 && Ds == 2) RII_QF3(float2, 2)
    else if (M == 16 && Ds == 2) RII_QF3(float2, 1)
    else return hipErrorInvalidValue;
#undef RII_QF3
#undef RII_QF
    return hipGetLastError();
}

bool lut_tile_supported(int M, int Ks, int Ds, int mx);
hipError_t launch_lut_tile_build_quant(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ds, float *d_lut,
                                       float *d_lohi, uint8_t *d_qlut, int32_t *d_slack, unsigned int *d_cand_cnt,
                                       uint32_t *d_gthr, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    const unsigned tiles = (unsigned) ((B + 15) / 16);
    if (Ds == 4)
        hipLaunchKernelGGL(lut_tile_build_kernel<float4>, dim3(tiles, M / 8, 4), dim3(256), 0, st, d_queries, B, d_codewords, M, d_lut,
                           reinterpret_cast<float2 *>(d_lohi));
    else
        hipLaunchKernelGGL(lut_tile_build_kernel<float2>, dim3(tiles, M / 8, 4), dim3(256), 0, st, d_queries, B, d_codewords, M, d_lut,
                           reinterpret_cast<float2 *>(d_lohi));
    if (M == 64)
        hipLaunchKernelGGL(qlut_tile_quant8_kernel, dim3((unsigned) ((B + 7) / 8), M / 32, 32), dim3(256), 0, st, d_lut,
                           reinterpret_cast<const float2 *>(d_lohi), B, M, d_qlut, d_slack, d_cand_cnt, d_gthr);
    else
        hipLaunchKernelGGL(qlut_tile_quant_kernel, dim3(tiles, M / 16, 16), dim3(256), 0, st, d_lut,
                           reinterpret_cast<const float2 *>(d_lohi), B, M, d_qlut, d_slack, d_cand_cnt, d_gthr);
    return hipGetLastError();
}

// fused: exact table in the plain [b][M*Ks] layout + quantisation
hipError_t launch_lut_build_quant(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ks, int Ds,
                                  int arch, float *d_lut, uint8_t *d_qc, uint8_t *d_qlut, int32_t *d_slack,
                                  unsigned int *d_cand_cnt, uint32_t *d_gthr, int mx, hipStream_t st, int levels)
{
    if (B == 0) return hipSuccess;
    if (levels != 63 && levels != 127 && levels != 255) return hipErrorInvalidValue;
    if (levels == 63 && Ds == 4 && M <= 32 && Ks <= 256) {   // Ds == 4: all three fvec_L2sqr variants coincide (rii_device.h)
        hipLaunchKernelGGL(lut_build_quant_regs_kernel, dim3((unsigned) B), dim3(256), 0, st, d_queries, B, d_codewords, M, Ks,
                           d_lut, d_qc, d_slack, d_cand_cnt, d_gthr);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
        return launch_qlut_interleave(d_qc, B, M, Ks, d_qlut, mx, st);
    }
    const size_t smem = (size_t) M * Ks * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lut_build_quant_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lut_build_quant_kernel, dim3((unsigned) B), dim3(256), smem, st, d_queries, B, d_codewords, M, Ks,
                       Ds, arch, d_lut, d_qc, d_slack, d_cand_cnt, d_gthr, levels);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_qlut_interleave(d_qc, B, M, Ks, d_qlut, mx, st);
}

// ---------------------------------------------------------------------------------------------------
// stage 1: quantised scan.  grid = (chunks, ceil(B/16)); 1024 threads; LDS = [M][Ks][16] u8 (+ thresholds).
//
// VALU budget is what bounds this kernel once the table entries are one byte (integer VALU ops issue at 4 cycles
// per wave64 on gfx950: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1 quad-cycle in profiles/), so the inner loop is
// built to spend ~1 VALU op per dword of table data: entries have kFsLevels = 63 levels, so kFsFlush = 4
// consecutive lookups can be added AS PACKED BYTES (4 x 63 < 256: no carry between the four queries of a dword)
// and only every 4th lookup is widened into the 16-bit pair accumulators.
// ---------------------------------------------------------------------------------------------------
struct FsArgs {
    const uint8_t *codes;
    int64_t n_codes;
    int M, Ks;
    const uint8_t *qlut;
    const int32_t *slack;
    int B;
    int64_t chunk_len;
    unsigned long long *cand;      // [B][cap] (a << 32 | local index)
    unsigned int *cand_count;      // [B]
    int cap;
    int lcap;                      // candidate slots per query staged in LDS by each block (0 = emit straight to global)
    uint16_t *segmin;              // MODE 1: [B][G] per-lane-segment minima of a(), G = gridDim.x * 1024
    const uint32_t *thr16;         // MODE 2: [B] fixed thresholds (candidate <=> a < thr16[b])
    uint32_t *gthr;                // MODE 0: [B] thresholds shared by all chunk-blocks of a tile (pre-set to 0xffff)
    int sample_stride;             // MODE 1: visit every sample_stride-th 1024-code slab of the chunk only (>= 1)
    int quarter = 0;               // fscan_mx_kernel: qlut holds quarter tables [tile][quarter][m][ks] u32 (qlut_fused_kernel)
    int dual = 0;                  // M = 16: two 16-query tiles per block (fscan_mx_dual_kernel)
    int bias = 0;                  // fscan_mx_*: initial value of the accumulators (128 M for tables of signed bytes = 255 levels, else 0)
    int pipe = 1;                  // fscan_mx_kernel MODE 0, unsigned table bytes: 1 = judge one group late (PIPE instances; engine option scan_pipe)
    FsTail tail;                   // fscan_mx_* MODE 0, TAIL instances: the last chunk-block of a tile re-ranks the tile's queries (round 4)
};

// TAIL instances of fscan_mx_*: the staged candidate records leave the CU write-through (agent-scope store = `sc1`) so that the
// fused re-rank -- another block of the SAME launch, possibly behind another XCD's L2 -- can read them with `sc1` loads.  (Records
// that missed the LDS staging area are stored plainly from the hot loop's rare branch -- an agent-scope store there costs the
// loop its register allocation -- and a block that had any is released with one L2 write-back before it arrives: fs_tail_arrive.)
__device__ __forceinline__ void fs_cand_store(unsigned long long *dst, unsigned long long rec)
{
    __hip_atomic_store(dst, rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MODE 0: top-1, thresholds adapt to the block's running minimum.  MODE 1 / 2: the two passes of top-k (k > 1):
// pass 1 records, per lane, the minimum a() over the codes that lane saw (a partition of the codes into G segments);
// the k-th smallest segment minimum v_k is >= the k-th smallest a() overall, so pass 2 keeps a() <= v_k + slack.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// widen the four byte-packed partial sums of a row into 8 registers of two u16 fields each:
// byte j of dword w is query 4w+j; even bytes -> acc[2w] (queries 4w, 4w+2), odd bytes -> acc[2w+1] (4w+1, 4w+3)
template <int QR> struct FsRow;                                   // one LDS row = QR one-byte entries
template <> struct FsRow<16> { typedef uint4 T; };
template <> struct FsRow<8> { typedef uint2 T; };

// odd bytes of a dword as two u16 fields, [b1, 0, b3, 0]: one v_perm_b32 (both sources are x, so the selector only
// ever names bytes of x; 0x0c selects the constant 0x00)
__device__ __forceinline__ uint32_t fs_odd_bytes(uint32_t x) { return __builtin_amdgcn_perm(x, x, 0x0c030c01u); }

template <int QR> __device__ __forceinline__ void fs_flush(uint32_t (&acc)[QR / 2], uint32_t (&pb)[QR / 4])
{
#pragma unroll
    for (int w = 0; w < QR / 4; ++w) {
        acc[2 * w] += pb[w] & 0x00ff00ffu;
        acc[2 * w + 1] += fs_odd_bytes(pb[w]);
        pb[w] = 0u;
    }
}

// byte J of a code word times the row size (1 << SH), i.e. the LDS offset of the row inside its subspace table, in ONE
// VALU instruction: the SDWA form of v_lshlrev_b32 selects and zero-extends the byte on the way in.
template <int J, int SH> __device__ __forceinline__ uint32_t fs_row_off(uint32_t w)
{
    static_assert(SH == 4 || SH == 3, "row size is 16 or 8 bytes");
    uint32_t r;
    if constexpr (SH == 4) {
        if constexpr (J == 0) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(w));
        if constexpr (J == 1) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(w));
        if constexpr (J == 2) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(w));
        if constexpr (J == 3) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(w));
    } else {
        if constexpr (J == 0) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(w));
        if constexpr (J == 1) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(w));
        if constexpr (J == 2) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(w));
        if constexpr (J == 3) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(w));
    }
    return r;
}

// The rows are fetched with hand-placed ds_read instructions: dynamic LDS starts at address 0 in this kernel (it has no
// static __shared__), so the row offset IS the address and the subspace table base rides in the instruction's 16-bit
// offset field; tables past 64 KiB need the upper part of their base added to the address (one more VALU instruction).
// The compiler does not count outstanding LDS operations issued from asm, so fs_wait4 names the registers that become
// valid: every use of a row is data-dependent on its wait.  (The compiler's own waits stay correct: LDS returns in
// order, extra outstanding operations only make its lgkmcnt conditions stricter.)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int QR> struct FsVec;
template <> struct FsVec<16> { typedef u32x4 T; };
template <> struct FsVec<8> { typedef u32x2 T; };

template <int QR, int KST, int M_, int J> __device__ __forceinline__ typename FsVec<QR>::T fs_row_issue(uint32_t w)
{
    constexpr int SH = QR == 16 ? 4 : 3;
    constexpr uint32_t tab = (uint32_t) M_ * KST * QR;
    constexpr uint32_t hi = tab & 0xffff0000u, lo = tab & 0xffffu;
    uint32_t addr = fs_row_off<J, SH>(w);
    if constexpr (hi != 0) addr += hi;
    typename FsVec<QR>::T r;
    if constexpr (QR == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(lo));
    else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(lo));
    return r;
}
// the four rows one code word selects (subspaces 4*I .. 4*I+3), in flight after this returns
template <int QR, int KST, int I> __device__ __forceinline__ void fs_word_issue(uint32_t w, typename FsVec<QR>::T (&r)[4])
{
    r[0] = fs_row_issue<QR, KST, 4 * I, 0>(w);
    r[1] = fs_row_issue<QR, KST, 4 * I + 1, 1>(w);
    r[2] = fs_row_issue<QR, KST, 4 * I + 2, 2>(w);
    r[3] = fs_row_issue<QR, KST, 4 * I + 3, 3>(w);
}
// rotated layout: lookup J (0/1) of a formatted code dword is a 16-bit (half, ks, slot) value; shifted by log2(row bytes)
// it IS the LDS address of the row (dynamic LDS starts at 0, the tables come first)
template <int QR, int J> __device__ __forceinline__ typename FsVec<QR>::T fs_rot_row_issue(uint32_t w)
{
    uint32_t addr;
    if constexpr (QR == 16) {
        if constexpr (J == 0) asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(addr) : "v"(w));
        else asm("v_lshlrev_b32_sdwa %0, 4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(addr) : "v"(w));
    } else {
        if constexpr (J == 0) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(addr) : "v"(w));
        else asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(addr) : "v"(w));
    }
    typename FsVec<QR>::T r;
    if constexpr (QR == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    else asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
// the four rows of lookups 4*I .. 4*I+3 (two formatted dwords), in flight after this returns.  ONE asm statement: the
// compiler pads every boundary between two asm statements with s_nop (it cannot see what they contain), and an s_nop costs
// an issue slot like any other instruction: 22 of them per code in the loop before this.
template <int QR> __device__ __forceinline__ void fs_rot_word_issue(uint32_t w0, uint32_t w1, typename FsVec<QR>::T (&r)[4])
{
    uint32_t a0, a1, a2, a3;
    if constexpr (QR == 16) {
        asm volatile("v_lshlrev_b32_sdwa %4, 4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                     "v_lshlrev_b32_sdwa %5, 4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                     "v_lshlrev_b32_sdwa %6, 4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                     "v_lshlrev_b32_sdwa %7, 4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                     "ds_read_b128 %0, %4\n\t"
                     "ds_read_b128 %1, %5\n\t"
                     "ds_read_b128 %2, %6\n\t"
                     "ds_read_b128 %3, %7"
                     : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                     : "v"(w0), "v"(w1));
    } else {
        asm volatile("v_lshlrev_b32_sdwa %4, 3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                     "v_lshlrev_b32_sdwa %5, 3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                     "v_lshlrev_b32_sdwa %6, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
                     "v_lshlrev_b32_sdwa %7, 3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                     "ds_read_b64 %0, %4\n\t"
                     "ds_read_b64 %1, %5\n\t"
                     "ds_read_b64 %2, %6\n\t"
                     "ds_read_b64 %3, %7"
                     : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                     : "v"(w0), "v"(w1));
    }
}
// wait until at most PENDING younger LDS operations are outstanding: the four rows named become valid
template <int PENDING, typename V> __device__ __forceinline__ void fs_wait4(V (&r)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(PENDING));
}

// a + b + c in one instruction.  Written as asm because the optimiser otherwise re-balances the integer sums below into
// pairwise adds (one more instruction per three-operand sum; the kernel is VALU-bound).
__device__ __forceinline__ uint32_t fs_add3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// byte-packed sum of four rows, <= 4 * kFsLevels per byte
template <int QR> __device__ __forceinline__ void fs_word_sum(const typename FsVec<QR>::T (&r)[4], uint32_t (&s)[QR / 4])
{
#pragma unroll
    for (int d = 0; d < QR / 4; ++d) s[d] = fs_add3(r[0][d], r[1][d], r[2][d]) + r[3][d];
}

// all MW groups of four lookups of one code against the byte tables: acc[] = 16-bit sums per query (packing: see
// fs_flush).  Two groups (8 rows) are consumed per step while the next two are already in flight.  ROT: w[] holds the
// formatted lookups of the rotated layout (two per dword, 2*MW dwords), else the plain code words (MW dwords).
template <int QR, int KST, int MW, int I, bool ROT, int NW>
__device__ __forceinline__ void fs_group_issue(const uint32_t (&w)[NW], typename FsVec<QR>::T (&r)[4])
{
    if constexpr (ROT) fs_rot_word_issue<QR>(w[2 * I], w[2 * I + 1], r);
    else fs_word_issue<QR, KST, I>(w[I], r);
}
template <int QR, int KST, int MW, int I, bool ROT, int NW>
__device__ __forceinline__ void fs_code_steps(const uint32_t (&w)[NW], uint32_t (&A)[QR / 4], uint32_t (&O)[QR / 4],
                                              typename FsVec<QR>::T (&ra)[4], typename FsVec<QR>::T (&rb)[4])
{
    if constexpr (I < MW) {
        typename FsVec<QR>::T na[4], nb[4];
        constexpr bool more = I + 2 < MW;
        if constexpr (more) {
            fs_group_issue<QR, KST, MW, I + 2, ROT, NW>(w, na);
            fs_group_issue<QR, KST, MW, I + 3, ROT, NW>(w, nb);
        }
        uint32_t sa[QR / 4], sb[QR / 4];
        fs_wait4<more ? 12 : 4>(ra);
        fs_word_sum<QR>(ra, sa);
        fs_wait4<more ? 8 : 0>(rb);
        fs_word_sum<QR>(rb, sb);
#pragma unroll
        for (int d = 0; d < QR / 4; ++d) {                 // A / O accumulators: see fs_rot_steps
            if constexpr (I == 0) {
                A[d] = sa[d] + sb[d];
                O[d] = fs_odd_bytes(sa[d]) + fs_odd_bytes(sb[d]);
            } else {
                A[d] = fs_add3(A[d], sa[d], sb[d]);
                O[d] = fs_add3(O[d], fs_odd_bytes(sa[d]), fs_odd_bytes(sb[d]));
            }
        }
        if constexpr (more) fs_code_steps<QR, KST, MW, I + 2, ROT, NW>(w, A, O, na, nb);
    }
}
template <int QR, int KST, int MW, bool ROT, int NW>
__device__ __forceinline__ void fs_code(const uint32_t (&w)[NW], uint32_t (&acc)[QR / 2])
{
    static_assert(MW % 2 == 0 && kFsFlush == 4, "pairs of lookup groups, 4 lookups per byte-packed sum");
    typename FsVec<QR>::T ra[4], rb[4];
    uint32_t A[QR / 4], O[QR / 4];
    fs_group_issue<QR, KST, MW, 0, ROT, NW>(w, ra);
    fs_group_issue<QR, KST, MW, 1, ROT, NW>(w, rb);
    fs_code_steps<QR, KST, MW, 0, ROT, NW>(w, A, O, ra, rb);
#pragma unroll
    for (int d = 0; d < QR / 4; ++d) {
        acc[2 * d] = A[d] - (O[d] << 8);
        acc[2 * d + 1] = O[d];
    }
}

// ---- rotated layout: the whole code in one go ----
// Byte-packed group sums s (four lookups, <= 252 per byte) feed TWO accumulators per dword column:
//   A += s          as a plain 32-bit integer (carries between the byte fields are allowed), and
//   O += odd bytes of s widened to two 16-bit fields (one v_perm_b32).
// With B0..B3 the true per-query sums (<= 32 * 63 < 2^11):  A = B0 + B1*2^8 + B2*2^16 + B3*2^24 (mod 2^32) and
// O = B1 + B3*2^16, hence  A - (O << 8) = B0 + B2*2^16  exactly: the even queries' 16-bit sums fall out of one shift and
// one subtraction per column at the end instead of an AND per group (8 VALU ops per 8 lookups and column instead of 10).
// The formatted lookups of the NEXT code this lane scans are fetched as soon as the last address of the current one has
// been formed (w[] is dead from there on): the loads travel under the remaining sums instead of stalling the loop head.
template <int QR, int MW, int I>
__device__ __forceinline__ void fs_rot_steps(uint32_t (&w)[2 * MW], uint32_t (&A)[QR / 4], uint32_t (&O)[QR / 4],
                                             typename FsVec<QR>::T (&ra)[4], typename FsVec<QR>::T (&rb)[4],
                                             const uint4 *__restrict__ next, bool prefetch)
{
    if constexpr (I < MW) {
        typename FsVec<QR>::T na[4], nb[4];
        constexpr bool more = I + 2 < MW;
        if constexpr (more) {
            fs_rot_word_issue<QR>(w[2 * (I + 2)], w[2 * (I + 2) + 1], na);
            fs_rot_word_issue<QR>(w[2 * (I + 3)], w[2 * (I + 3) + 1], nb);
        }
        if constexpr (I + 4 == MW || (MW == 2 && I == 0)) {
            if (prefetch) {
#pragma unroll
                for (int i = 0; i < MW / 2; ++i) {
                    const uint4 v = next[i];
                    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
                }
            }
        }
        uint32_t sa[QR / 4], sb[QR / 4];
        fs_wait4<more ? 12 : 4>(ra);
        fs_word_sum<QR>(ra, sa);
        fs_wait4<more ? 8 : 0>(rb);
        fs_word_sum<QR>(rb, sb);
#pragma unroll
        for (int d = 0; d < QR / 4; ++d) {
            if constexpr (I == 0) {
                A[d] = sa[d] + sb[d];
                O[d] = fs_odd_bytes(sa[d]) + fs_odd_bytes(sb[d]);
            } else {
                A[d] = fs_add3(A[d], sa[d], sb[d]);
                O[d] = fs_add3(O[d], fs_odd_bytes(sa[d]), fs_odd_bytes(sb[d]));
            }
        }
        if constexpr (more) fs_rot_steps<QR, MW, I + 2>(w, A, O, na, nb, next, prefetch);
    }
}
template <int QR, int MW>
__device__ __forceinline__ void fs_rot_code(uint32_t (&w)[2 * MW], uint32_t (&acc)[QR / 2], const uint4 *__restrict__ next,
                                            bool prefetch)
{
    static_assert(MW % 2 == 0 && MW >= 4 && kFsFlush == 4, "pairs of lookup groups, 4 lookups per byte-packed sum");
    typename FsVec<QR>::T ra[4], rb[4];
    uint32_t A[QR / 4], O[QR / 4];
    fs_rot_word_issue<QR>(w[0], w[1], ra);
    fs_rot_word_issue<QR>(w[2], w[3], rb);
    fs_rot_steps<QR, MW, 0>(w, A, O, ra, rb, next, prefetch);
#pragma unroll
    for (int d = 0; d < QR / 4; ++d) {
        acc[2 * d] = A[d] - (O[d] << 8);
        acc[2 * d + 1] = O[d];
    }
}

__device__ __forceinline__ void fs_add(uint32_t (&pb)[4], const uint4 &v)
{
    pb[0] += v.x; pb[1] += v.y; pb[2] += v.z; pb[3] += v.w;
}
__device__ __forceinline__ void fs_add(uint32_t (&pb)[2], const uint2 &v)
{
    pb[0] += v.x; pb[1] += v.y;
}
template <int NR> __device__ __forceinline__ uint32_t fs_get(const uint32_t (&acc)[NR], int q)
{
    const int w = q >> 2, j = q & 3;
    const uint32_t r = acc[2 * w + (j & 1)];
    return (j & 2) ? (r >> 16) : (r & 0xffffu);
}
// thresholds live in LDS as 8 words in the SAME packing as acc[]: field = (running minimum + slack + 1), <= 0xffff.
// Lowering one 16-bit field is a rare event: CAS loop on the containing word.
__device__ __forceinline__ void fs_thr_lower(uint32_t *word, int high, uint32_t v16)
{
    uint32_t old = *reinterpret_cast<volatile uint32_t *>(word);
    for (;;) {
        const uint32_t cur = high ? (old >> 16) : (old & 0xffffu);
        if (v16 >= cur) return;
        const uint32_t nw = high ? ((old & 0xffffu) | (v16 << 16)) : ((old & 0xffff0000u) | v16);
        const uint32_t prev = atomicCAS(word, old, nw);
        if (prev == old) return;
        old = prev;
    }
}
__device__ __forceinline__ uint32_t fs_thr_of(uint32_t a, uint32_t slack)
{
    const uint32_t t = a + slack + 1u;
    return t > 0xffffu ? 0xffffu : t;
}

template <int MW, int KST, int MODE, int QR, bool ROT = false>
__global__ __launch_bounds__(kFsThreads) void fscan_kernel(FsArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = MW ? MW * 4 : p.M;
    const int Ks = KST ? KST : p.Ks;
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    const size_t lut_bytes = (size_t) M * Ks * QR;
    uint32_t *s_thr = reinterpret_cast<uint32_t *>(smem + lut_bytes);          // [QR/2] packed thresholds
    // candidates are staged per block: LDS counter + list per query, appended to the global lists with ONE atomicAdd per
    // (block, query) at the end -- hundreds of chunk-blocks bumping the same global counter once per candidate was the
    // floor of the small-batch scan.  A full local list spills straight to the global one.
    uint32_t *s_lcnt = reinterpret_cast<uint32_t *>(smem + lut_bytes + 64);     // [QR] staged candidates, then [QR] global bases
    unsigned long long *s_lcand = reinterpret_cast<unsigned long long *>(smem + lut_bytes + 64 + QR * 8);   // [QR][lcap]
    typedef typename FsRow<QR>::T Row;
    {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(p.qlut + (size_t) tile * lut_bytes);
        uint4 *d4 = reinterpret_cast<uint4 *>(smem);
        for (size_t i = tid; i < lut_bytes / 16; i += kFsThreads) d4[i] = s4[i];
        if (tid < QR / 2) {
            // word i = 2w + parity: low = query 4w+parity, high = query 4w+parity+2.  Queries past the end of the batch
            // (last tile) get threshold 0: they can never "hit", or every code of the chunk would take the slow path
            const int q0 = 4 * (tid >> 1) + (tid & 1), b0 = tile * QR + q0, b1 = b0 + 2;
            uint32_t lo = b0 < p.B ? 0xffffu : 0u, hi = b1 < p.B ? 0xffffu : 0u;
            if constexpr (MODE == 2) {
                lo = b0 < p.B ? p.thr16[b0] : 0u;
                hi = b1 < p.B ? p.thr16[b1] : 0u;
            }
            s_thr[tid] = (lo & 0xffffu) | (hi << 16);
        }
        if (tid < 2 * QR) s_lcnt[tid] = 0u;
    }
    __syncthreads();
    const Row *lut = reinterpret_cast<const Row *>(smem);
    uint32_t smin[QR / 2];
#pragma unroll
    for (int i = 0; i < QR / 2; ++i) smin[i] = 0xffffffffu;

    const int64_t c_begin = (int64_t) blockIdx.x * p.chunk_len;
    int64_t c_end = c_begin + p.chunk_len;
    if (c_end > p.n_codes) c_end = p.n_codes;
    const int64_t span = c_end > c_begin ? c_end - c_begin : 0;
    const int iters = (int) ((span + kFsThreads - 1) / kFsThreads);

    uint32_t acc0[QR / 2] = {};
    int64_t n0 = 0;
    bool active0 = false;
    // candidate test of one code (its 16-bit sums in a[], scan position n) against the thresholds in LDS
    auto test_and_emit = [&](const uint32_t (&acc)[QR / 2], int64_t n) {
        // candidate test on the packed pairs: sat(thr+1 - a) != 0  <=>  a <= thr
        uint32_t thr[QR / 2];
#pragma unroll
        for (int i = 0; i < QR / 8; ++i) {
            const uint4 t4 = reinterpret_cast<const uint4 *>(s_thr)[i];
            thr[4 * i] = t4.x; thr[4 * i + 1] = t4.y; thr[4 * i + 2] = t4.z; thr[4 * i + 3] = t4.w;
        }
        uint32_t hit = 0u, dsat[QR / 2];
#pragma unroll
        for (int i = 0; i < QR / 2; ++i) {
            const u16x2 d = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, thr[i]),
                                                          __builtin_bit_cast(u16x2, acc[i]));
            dsat[i] = __builtin_bit_cast(uint32_t, d);
            hit |= dsat[i];
        }
        if (hit) {
            // rare path: walk only the queries whose field is non-zero (usually exactly one)
            uint32_t mask = 0u;
#pragma unroll
            for (int i = 0; i < QR / 2; ++i) {        // register i: low field = query 4(i/2)+(i&1), high field = +2
                const int ql = 4 * (i >> 1) + (i & 1);
                mask |= ((dsat[i] & 0xffffu) ? 1u : 0u) << ql;
                mask |= ((dsat[i] >> 16) ? 1u : 0u) << (ql + 2);
            }
            while (mask) {
                const int q = __ffs((int) mask) - 1;
                mask &= mask - 1u;
                const int b = tile * QR + q;
                const int reg = 2 * (q >> 2) + (q & 1), high = (q >> 1) & 1;
                uint32_t av = 0u, tv = 0u;
#pragma unroll
                for (int i = 0; i < QR / 2; ++i)          // register select without dynamic indexing
                    if (i == reg) { av = acc[i]; tv = thr[i]; }
                const uint32_t a = high ? (av >> 16) : (av & 0xffffu);
                const uint32_t t = high ? (tv >> 16) : (tv & 0xffffu);
                if (a < t && b < p.B) {
                    if constexpr (MODE == 0) {
                        const uint32_t nt = fs_thr_of(a, (uint32_t) p.slack[b]);
                        if (nt < t) {
                            fs_thr_lower(&s_thr[reg], high, nt);
                            atomicMin(&p.gthr[b], nt);       // let the other chunks of this tile prune with it too
                        }
                    }
                    const unsigned long long rec = ((unsigned long long) a << 32) | (uint32_t) n;
                    bool staged = false;
                    if (p.lcap > 0) {
                        const unsigned int lp = atomicAdd(&s_lcnt[q], 1u);
                        if (lp < (unsigned int) p.lcap) { s_lcand[(size_t) q * p.lcap + lp] = rec; staged = true; }
                    }
                    if (!staged) {
                        const unsigned int pos = atomicAdd(&p.cand_count[b], 1u);
                        if (pos < (unsigned int) p.cap) p.cand[(size_t) b * p.cap + pos] = rec;
                    }
                }
            }
        }
    };
    // MODE 1 only needs an UPPER bound on the k-th smallest sum, so it may look at a strided sample of the codes
    const int it_step = (MODE == 1) ? p.sample_stride : 1;
    // rotated layout: the formatted lookups of a lane's code, fetched one trip ahead (fs_rot_code)
    uint32_t wrot[(ROT && MW) ? 2 * MW : 1];
    if constexpr (ROT && MW != 0) {
        const int64_t nf = c_begin + tid;
        if (nf < c_end) {
            const uint4 *cp = reinterpret_cast<const uint4 *>(p.codes + (size_t) nf * (MW * 8));
#pragma unroll
            for (int i = 0; i < MW / 2; ++i) {
                const uint4 v = cp[i];
                wrot[4 * i] = v.x; wrot[4 * i + 1] = v.y; wrot[4 * i + 2] = v.z; wrot[4 * i + 3] = v.w;
            }
        }
    }
    for (int it = 0; it < iters; it += it_step) {
        const int64_t n = c_begin + (int64_t) it * kFsThreads + tid;
        const bool active = n < c_end;
        uint32_t acc[QR / 2], pb[QR / 4];
#pragma unroll
        for (int i = 0; i < QR / 2; ++i) acc[i] = 0u;
#pragma unroll
        for (int i = 0; i < QR / 4; ++i) pb[i] = 0u;
        if (active) {
            if constexpr (MW != 0 && ROT) {
                // formatted lookups: 2 bytes each, 8 * MW bytes per code
                const int64_t nn = n + (int64_t) it_step * kFsThreads;
                fs_rot_code<QR, MW>(wrot, acc, reinterpret_cast<const uint4 *>(p.codes + (size_t) nn * (MW * 8)), nn < c_end);
            } else if constexpr (MW != 0) {
                const uint8_t *cp = p.codes + (size_t) n * (MW * 4);
                uint32_t w[MW ? MW : 1];
                if constexpr (MW % 4 == 0) {
#pragma unroll
                    for (int i = 0; i < MW / 4; ++i) {
                        const uint4 v = reinterpret_cast<const uint4 *>(cp)[i];
                        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < MW / 2; ++i) {
                        const uint2 v = reinterpret_cast<const uint2 *>(cp)[i];
                        w[2 * i] = v.x; w[2 * i + 1] = v.y;
                    }
                }
                if constexpr (kFsFlush == 4 && MW % 2 == 0) {
                    fs_code<QR, KST, MW, false, (MW ? MW : 1)>(w, acc);
                } else {
#pragma unroll
                    for (int i = 0; i < MW; ++i) {          // 4 lookups per code word; flush every kFsFlush lookups
#pragma unroll
                        for (int j = 0; j < 4; ++j) fs_add(pb, lut[(i * 4 + j) * KST + ((w[i] >> (8 * j)) & 0xffu)]);
                        if ((i * 4 + 4) % kFsFlush == 0 || i == MW - 1) fs_flush<QR>(acc, pb);
                    }
                }
            } else {
                const uint8_t *c = p.codes + (size_t) n * M;
                int pend = 0;
                for (int m = 0; m < M; ++m) {
                    fs_add(pb, lut[m * Ks + c[m]]);
                    if (++pend == kFsFlush) { fs_flush<QR>(acc, pb); pend = 0; }
                }
                fs_flush<QR>(acc, pb);
            }
        }
        if constexpr (MODE == 1) {
            if (active) {
#pragma unroll
                for (int i = 0; i < QR / 2; ++i)
                    smin[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, smin[i]),
                                                                                    __builtin_bit_cast(u16x2, acc[i])));
            }
            continue;
        }
        if (MODE == 0 && it == 0) {
            // warm-up: publish thresholds from the first <=1024 codes before anybody tests candidacy
#pragma unroll
            for (int q = 0; q < QR; ++q) {
                const int b = tile * QR + q;
                uint32_t t = active ? fs_get(acc, q) : 0xffffffffu;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t o = __shfl_xor(t, off);
                    t = o < t ? o : t;
                }
                if ((tid & 63) == 0 && t != 0xffffffffu && b < p.B) {
                    const uint32_t nt = fs_thr_of(t, (uint32_t) p.slack[b]);
                    fs_thr_lower(&s_thr[2 * (q >> 2) + (q & 1)], (q >> 1) & 1, nt);
                }
            }
            __syncthreads();
        }
        if (MODE == 0 && tid < QR) {
            // adopt thresholds published by the blocks scanning the other chunks for the same queries.  A stale value
            // only means a few more candidates: thresholds are upper bounds on (global minimum + slack + 1) at all times.
            const int b = tile * QR + tid;
            if (b < p.B) {
                const uint32_t g = __hip_atomic_load(&p.gthr[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (it == 0) {
                    // publish the block's warm-up minimum ONCE (per-wave atomics from hundreds of chunk-blocks on the same
                    // 16 words used to cost more than the scan itself at small batches)
                    const uint32_t word = s_thr[2 * (tid >> 2) + (tid & 1)];
                    const uint32_t mine = ((tid >> 1) & 1) ? (word >> 16) : (word & 0xffffu);
                    if (mine < g) atomicMin(&p.gthr[b], mine);
                }
                fs_thr_lower(&s_thr[2 * (tid >> 2) + (tid & 1)], (tid >> 1) & 1, g);
            }
        }
        if (MODE == 0 && it == 0) {
            // the first slab is judged LAST: right now the thresholds only know this block's first 1024 codes, at the end
            // they know the whole chunk and what the other chunks published (short chunks used to emit most of their
            // candidates here)
#pragma unroll
            for (int i = 0; i < QR / 2; ++i) acc0[i] = acc[i];
            n0 = n;
            active0 = active;
        } else if (active) {
            test_and_emit(acc, n);
        }
    }
    if (MODE == 0 && active0) test_and_emit(acc0, n0);
    if (MODE == 0 && p.lcap > 0) {
        __syncthreads();
        if (tid < QR) {
            const int b = tile * QR + tid;
            const unsigned int c = min(s_lcnt[tid], (unsigned int) p.lcap);
            s_lcnt[QR + tid] = (c && b < p.B) ? atomicAdd(&p.cand_count[b], c) : 0u;
        }
        __syncthreads();
        for (int q = tid >> 6; q < QR; q += kFsThreads >> 6) {        // one wave per query
            const int b = tile * QR + q;
            const unsigned int c = min(s_lcnt[q], (unsigned int) p.lcap), base = s_lcnt[QR + q];
            for (unsigned int i = tid & 63; i < c; i += 64)
                if (base + i < (unsigned int) p.cap) p.cand[(size_t) b * p.cap + base + i] = s_lcand[(size_t) q * p.lcap + i];
        }
    }
    if constexpr (MODE == 1) {
        const size_t G = (size_t) gridDim.x * kFsThreads;
        const size_t seg = (size_t) blockIdx.x * kFsThreads + tid;
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            const int b = tile * QR + q;
            if (b < p.B) p.segmin[(size_t) b * G + seg] = (uint16_t) fs_get(smin, q);
        }
    }
}

template <int MW, int KST, int MODE, int QR, bool ROT = false>
static hipError_t launch_fscan_t(const FsArgs &a, int chunks, int tiles, hipStream_t st)
{
    const size_t tab = (size_t) a.M * a.Ks * QR + 64 + (size_t) QR * 8;
    FsArgs b = a;
    // top-1 only: the fixed-threshold pass of top-k emits so often that LDS atomics queueing behind the table reads cost
    // more than the global ones they save (measured)
    b.lcap = (MODE != 0) ? 0 : (int) std::min<size_t>(128, (kFsLdsBytes - tab) / ((size_t) QR * 8));
    const size_t smem = tab + (size_t) QR * 8 * b.lcap;
    auto kern = fscan_kernel<MW, KST, MODE, QR, ROT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    launch_timed(kern, dim3(chunks, tiles), dim3(kFsThreads), smem, st, b);
    return hipGetLastError();
}

// queries per LDS row: 16 (ds_read_b128) while the byte tables of 16 queries fit, else 8 (ds_read_b64), else 0
int fastscan_rows(int M, int Ks)
{
    if (M > 256) return 0;
    if ((size_t) M * Ks * 16 + 64 <= (size_t) kMaxLutLdsBytes) return 16;
    if ((size_t) M * Ks * 8 + 64 <= (size_t) kMaxLutLdsBytes) return 8;
    return 0;
}
bool fastscan_supported(int M, int Ks) { return fastscan_rows(M, Ks) != 0; }
int fastscan_max_sum(int M, int levels) { return M * (levels > 0 ? levels : kFsLevels); }

// shapes with the conflict-free rotated layout: whole groups of G = 16 subspaces (the lanes of a ds_read_b128 service
// group) and Ks = 256 (the (half, ks, slot) lookup value fits 16 bits)
bool lut_tile_supported(int M, int Ks, int Ds, int mx) { return (Ds == 4 || Ds == 2) && fs_rot_supported(M, Ks, mx); }

bool fs_rot_supported(int M, int Ks, int mx)
{
    const int qr = fastscan_rows(M, Ks);
    if (Ks != 256) return false;
    if (qr == 16 && (M == 16 || M == 32)) return true;
    // M = 64 (8-byte rows): only fscan_mx_kernel has a rotated form (as 16-bit lookups for fscan_kernel it was tried and lost:
    // 128 formatted bytes per code and lane make that loop load-bound, 1.9 ms against 1.3 ms with the plain codes in scan order)
    return mx && qr == 8 && M == 64;
}

// =====================================================================================================================
// fscan_mx_kernel: the filter scan of the rotated shapes (Ks = 256, 16-byte rows, M = 16 / 32) with the byte sums taken by
// the matrix cores.
//
// fscan_kernel above spends 128 of its ~207 VALU instructions per 64 codes on adding table bytes; it is bound by VALU issue
// with the LDS half idle.  v_smfmac_i32_16x16x128_i8 can do those additions: its dense operand takes 32 bytes per lane --
// TWO table rows as the ds_read_b128 delivered them -- and with the sparse operand set to a one-hot pattern
//     A[i][k] = 1  <=>  k is byte i or byte 16 + i of a lane's 32          (row i of the output = query i of the tile)
// the instruction computes  D[i][n] = sum over the four lanes 16 g + n (g = 0..3) of (row0[i] + row1[i]):  the eight rows
// four lanes fetched for code n, summed per query, as 32-bit integers, already transposed (lane 16 (i / 4) + n holds queries
// 4 (i / 4) .. + 3 of code n in its four accumulator registers).  No byte packing, no carries, no widening: M / 8 matrix
// instructions per 16 codes replace all the additions.  (Operand semantics probed on the device: tools/ubench/
// smfmac_probe.hip, mfma_reduce.hip.  This is not a GEMM in disguise: 1/16 of the multiplier array does useful work, the
// instruction is used as a 128-input adder tree with a free transpose.)
//
// Work split: a wave owns groups of 16 consecutive codes; lane (g, n) = 16 g + n fetches M / 4 of code n's M rows.  Which
// ones is chosen so that every ds_read_b128 service group of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32) meets 16
// different bank slots whatever the data: with the rotated table layout (slot = subspace mod 16) lookup t of lane (g, n) is
//     subspace 16 (t / 4) + (n + 4 (t mod 4) + e(g, n)) mod 16,   e = (g & 1) ^ [n in 4..11]  +  2 (g >> 1),
// (per code the four lanes cover offsets e = 0..3 once each; inside a service group the lanes of one g form either
// {0-3,12-15} or {4-11} and a common shift keeps 16 columns on 16 slots).  The order is baked into a permuted copy of the code
// bytes (fcodes_mx_format_kernel): per group 64 lanes x M / 4 bytes, one coalesced 512-byte (M = 32) load per wave; the subspace
// of a byte follows from its position, so the (half, slot) part of a row's LDS address is a per-lane constant.
// =====================================================================================================================
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));

// M = 64 (8-byte rows, ds_read_b64: service groups {0-31}, {32-63}, 32 bank slots): lookup t of lane (g, n) is subspace
//     32 (g >> 1) + (n + 16 (g & 1) + t) mod 32,   t = 0 .. 15
// (the two lanes of a half cover the 32 offsets once; a service group holds g = 0, 1 or g = 2, 3 with all 16 columns).
__host__ __device__ __forceinline__ int fs_mx_subspace64(int g, int col, int t) { return 32 * (g >> 1) + ((col + 16 * (g & 1) + t) & 31); }
__host__ __device__ __forceinline__ int fs_mx_subspace(int g, int col, int t)
{
    const int in_mid = (col >= 4 && col < 12) ? 1 : 0;
    const int e = ((g & 1) ^ in_mid) + 2 * (g >> 1);
    return 16 * (t >> 2) + ((col + 4 * (t & 3) + e) & 15);
}

// codes [n][M] u8 -> [group = n / 16][lane = 16 g + n % 16][t] u8: the code bytes in the order the lanes consume them (which
// subspace a byte belongs to follows from its position: fs_mx_subspace); positions in [n1, end of the last group) are
// filled with 0 (their sums are never judged).  n0 must be a multiple of 16.
__global__ __launch_bounds__(256) void fcodes_mx_format_kernel(const uint8_t *__restrict__ codes, const int64_t *__restrict__ ids,
                                                               int64_t n0, int64_t n1, int M, uint8_t *__restrict__ out)
{
    // one thread per output DWORD (four consecutive lookups of a lane): one id load and four bytes of one code row per thread,
    // coalesced 4-byte stores (a thread per byte cost 13 us per 100 k gathered codes: subset search pays this per batch)
    const int T = M / 4, TW = T / 4;                        // lookups / dwords per lane and group
    const int64_t n1p = (n1 + 15) / 16 * 16;
    const int64_t total = (n1p - n0) * M / 4;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out + (size_t) n0 * M);
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int u = (int) (i % TW);                       // dword u of the lane: lookups 4u .. 4u + 3
        const int64_t gl = i / TW;
        const int lane = (int) (gl & 63);
        const int64_t n = n0 + (gl >> 6) * 16 + (lane & 15);
        uint32_t w = 0u;
        if (n < n1) {
            const uint8_t *row = codes + (size_t) (ids ? ids[n] : n) * M;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * u + j;
                const int m = (M == 64) ? fs_mx_subspace64(lane >> 4, lane & 15, t) : fs_mx_subspace(lane >> 4, lane & 15, t);
                w |= (uint32_t) row[m] << (8 * j);
            }
        }
        out32[i] = w;
    }
}

// M = 16 / 32 (round 4): one thread per (group, lane) writes the lane's M / 4 lookups at once.  Inside a 16-byte half of the code row
// the lane's four subspaces are (col + e + 4 j) & 15, j = 0 .. 3: byte r = (col + e) & 3 of the four dwords, taken in rotated dword
// order starting at dword d0 = ((col + e) >> 2) & 3 -- one 16-byte load, four byte extracts, one pack and one v_alignbyte per output
// dword instead of four single-byte loads (subset search pays this kernel per batch: 7.7 us per 100 k gathered codes before).
template <int M_>
__global__ __launch_bounds__(256) void fcodes_mx_format_rows_kernel(const uint8_t *__restrict__ codes, const int64_t *__restrict__ ids, int64_t n0,
                                                                    int64_t n1, uint8_t *__restrict__ out)
{
    constexpr int H = M_ / 16;                                 // 16-byte halves of a code row = output dwords per lane
    const int64_t n1p = (n1 + 15) / 16 * 16;
    const int64_t total = (n1p - n0) * 4;                      // lane slots: 64 per group of 16 codes
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out + (size_t) n0 * M_);
    for (int64_t gl = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; gl < total; gl += (int64_t) gridDim.x * blockDim.x) {
        const int lane = (int) (gl & 63), g = lane >> 4, col = lane & 15;
        const int64_t n = n0 + (gl >> 6) * 16 + col;
        uint32_t w[H];
#pragma unroll
        for (int h = 0; h < H; ++h) w[h] = 0u;
        if (n < n1) {
            const uint4 *row = reinterpret_cast<const uint4 *>(codes + (size_t) (ids ? ids[n] : n) * M_);
            const int in_mid = (col >= 4 && col < 12) ? 1 : 0;
            const int ce = col + (((g & 1) ^ in_mid) + 2 * (g >> 1));      // col + e(g, col)
            const uint32_t r8 = (uint32_t) (ce & 3) * 8u, d0 = (uint32_t) ((ce >> 2) & 3);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const uint4 v = row[h];
                const uint32_t pk = ((v.x >> r8) & 0xffu) | (((v.y >> r8) & 0xffu) << 8) | (((v.z >> r8) & 0xffu) << 16) | (((v.w >> r8) & 0xffu) << 24);
                w[h] = __builtin_amdgcn_alignbyte(pk, pk, d0);              // bytes [d0, d0 + 1, d0 + 2, d0 + 3] (mod 4) of pk
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) out32[gl * H + h] = w[h];
    }
}

// the compressed one-hot operand for output row i (see the header comment): stored bytes 2 (i / 4) and 8 + 2 (i / 4) are 1 and
// carry the 2-bit index i % 4; the other stored bytes are 0 (their indices only have to differ from their pair's)
__device__ __forceinline__ void fs_mx_pattern(int i, v4i_t &a, int &idx)
{
    const int s1 = 2 * (i >> 2), s2 = 8 + s1;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    uint32_t x = 0u;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        const bool one = (f == s1 || f == s2), pair = ((f ^ 1) == s1 || (f ^ 1) == s2);
        if (one) w[f >> 2] |= 1u << (8 * (f & 3));
        const uint32_t v = one ? (uint32_t) (i & 3) : pair ? (uint32_t) ((i + 2) & 3) : (uint32_t) (f & 1);
        x |= v << (2 * f);
    }
    a = v4i_t{(int) w[0], (int) w[1], (int) w[2], (int) w[3]};
    idx = (int) x;
}

typedef const __attribute__((address_space(3))) v4i_t *fs_lds_row_t;
// A fresh accumulator for the sparse matrix instruction (it accumulates in place: D = A x B + D).  Round 5: two v_pk_mov_b32 (one
// 64-bit register pair each) instead of the four v_mov_b32 the compiler writes for `acc = zero` -- per group of 16 codes that is
// 2 of 14 (M = 32) / 4 of 20 (two M = 16 tiles) vector instructions fewer on an issue port that is co-limiting with the LDS.
#ifndef RII_PK_ZERO
#define RII_PK_ZERO 1
#endif
typedef int fs_v2i_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4i_t fs_acc_init(const v4i_t &zero)
{
#if RII_PK_ZERO
    const fs_v2i_t zl = __builtin_shufflevector(zero, zero, 0, 1), zh = __builtin_shufflevector(zero, zero, 2, 3);
    fs_v2i_t lo, hi;
    asm volatile("v_pk_mov_b32 %0, %2, %2 op_sel:[0,1]\n\tv_pk_mov_b32 %1, %3, %3 op_sel:[0,1]" : "=v"(lo), "=v"(hi) : "v"(zl), "v"(zh));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
#else
    return zero;
#endif
}
// the same from the inline constant 0 (tables with unsigned bytes: no bias, no zero registers kept)
__device__ __forceinline__ v4i_t fs_acc_init0()
{
#if RII_PK_ZERO
    fs_v2i_t lo, hi;
    asm volatile("v_pk_mov_b32 %0, 0, 0\n\tv_pk_mov_b32 %1, 0, 0" : "=v"(lo), "=v"(hi));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
#else
    return v4i_t{0, 0, 0, 0};
#endif
}
// The lookups are the code bytes themselves (one byte per row to fetch).  The LDS address of the row of lookup t is
//     (half << 16) | (ks << 8) | (slot << 4)          (rotated layout: row = half * 4096 + ks * 16 + slot, 16 bytes each;
// dynamic LDS starts at address 0 in this kernel); half and slot depend on the lane and on t only, so they sit in T registers per
// lane (fs_mx_consts) and ONE v_perm_b32 splices the code byte in as byte 1.
template <int T> __device__ __forceinline__ void fs_mx_consts(int lane, uint32_t (&C)[T])
{
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int m = fs_mx_subspace(lane >> 4, lane & 15, t);
        C[t] = ((uint32_t) (m >> 4) << 16) | ((uint32_t) (m & 15) << 4);
    }
}
// selector of v_perm_b32(C, w, sel): bytes [C.0, w.J, C.2, C.3]  ({C, w} are bytes {4..7, 0..3} of the instruction's source pair)
__device__ __forceinline__ constexpr uint32_t fs_mx_sel(int J) { return 0x07060004u | ((uint32_t) J << 8); }
template <int T> struct FsMxW;                                      // the T lookups (bytes) of a lane for one group
template <> struct FsMxW<8> { typedef u32x2 V; };
template <> struct FsMxW<4> { typedef uint32_t V; };
template <> struct FsMxW<16> { typedef u32x4 V; };
template <int I> __device__ __forceinline__ uint32_t fs_mx_dword(const u32x2 &w) { return I == 0 ? w.x : w.y; }
template <int I> __device__ __forceinline__ uint32_t fs_mx_dword(const uint32_t &w) { return w; }
// lookups of a later trip, fetched from asm (the compiler would wait for them with vmcnt(0), i.e. for the load it issued a moment
// ago): four loads are in flight per wave, every use has exactly three younger ones behind it
template <int OFF> __device__ __forceinline__ void fs_mx_load(u32x2 &q, const u32x2 *p) { asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(q) : "v"(p), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void fs_mx_load(uint32_t &q, const uint32_t *p) { asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(q) : "v"(p), "n"(OFF)); }
template <int N, typename V> __device__ __forceinline__ void fs_mx_vmwait(V &q) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q) : "n"(N)); }

// compiler-scheduled form (warm-up and tail groups)
template <int T> __device__ __forceinline__ void fs_mx_issue(const typename FsMxW<T>::V &w, const uint32_t (&C)[T], v4i_t (&r)[T])
{
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t wd = (t < 4) ? fs_mx_dword<0>(w) : fs_mx_dword<1>(w);
        const uint32_t addr = __builtin_amdgcn_perm(C[t], wd, fs_mx_sel(t & 3));
        r[t] = *(fs_lds_row_t) (uintptr_t) addr;
    }
}
// The same with hand-placed ds_read instructions and waits (the hot loop).  The compiler's own waitcnt bookkeeping loses
// track of which loads are outstanding across the (rare) candidate branch and the loop edge and falls back to lgkmcnt(0) in
// front of every group -- draining the reads issued for the NEXT groups.  Issued from asm the reads are invisible to it;
// fs_mx_wait names the registers that become valid (LDS returns in order, so the compiler's own waits for its own LDS
// operations only get stricter).
// two rows (lookups J, J + 1 of dword w): what one matrix instruction consumes
template <int J> __device__ __forceinline__ void fs_mx_issue2(uint32_t w, uint32_t c0, uint32_t c1, v4i_t &r0, v4i_t &r1)
{
    uint32_t a0, a1;
    asm volatile("v_perm_b32 %2, %4, %6, %7\n\t"
                 "v_perm_b32 %3, %5, %6, %8\n\t"
                 "ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %3"
                 : "=v"(r0), "=v"(r1), "=&v"(a0), "=&v"(a1)
                 : "v"(c0), "v"(c1), "v"(w), "s"(fs_mx_sel(J)), "s"(fs_mx_sel(J + 1)));
}
template <int T> __device__ __forceinline__ void fs_mx_issue_hot(const typename FsMxW<T>::V &w, const uint32_t (&C)[T], v4i_t (&r)[T])
{
    fs_mx_issue2<0>(fs_mx_dword<0>(w), C[0], C[1], r[0], r[1]);
    fs_mx_issue2<2>(fs_mx_dword<0>(w), C[2], C[3], r[2], r[3]);
    if constexpr (T == 8) {
        fs_mx_issue2<0>(fs_mx_dword<1>(w), C[4], C[5], r[4], r[5]);
        fs_mx_issue2<2>(fs_mx_dword<1>(w), C[6], C[7], r[6], r[7]);
    }
}
// one group: the rows in r[] (already waited for) go through the matrix core pair by pair, and as soon as a pair has been
// issued its registers take the rows of the group two ahead (lookups wn) -- the reads travel under the remaining matrix
// instructions and the whole next group
template <int T, bool Z0 = false> __device__ __forceinline__ v4i_t fs_mx_reduce_refill(v4i_t (&r)[T], const typename FsMxW<T>::V &wn, const uint32_t (&C)[T],
                                                                      const v4i_t &spa, int spidx, const v4i_t &zero)
{
    v4i_t acc = Z0 ? fs_acc_init0() : fs_acc_init(zero);
    {
        const v8i_t b = __builtin_shufflevector(r[0], r[1], 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
        fs_mx_issue2<0>(fs_mx_dword<0>(wn), C[0], C[1], r[0], r[1]);
    }
    {
        const v8i_t b = __builtin_shufflevector(r[2], r[3], 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
        fs_mx_issue2<2>(fs_mx_dword<0>(wn), C[2], C[3], r[2], r[3]);
    }
    if constexpr (T == 8) {
        {
            const v8i_t b = __builtin_shufflevector(r[4], r[5], 0, 1, 2, 3, 4, 5, 6, 7);
            acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
            fs_mx_issue2<0>(fs_mx_dword<1>(wn), C[4], C[5], r[4], r[5]);
        }
        {
            const v8i_t b = __builtin_shufflevector(r[6], r[7], 0, 1, 2, 3, 4, 5, 6, 7);
            acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
            fs_mx_issue2<2>(fs_mx_dword<1>(wn), C[6], C[7], r[6], r[7]);
        }
    }
    return acc;
}
// at most PENDING younger LDS operations stay outstanding: the rows named are valid after this
template <int PENDING> __device__ __forceinline__ void fs_mx_wait(v4i_t (&r)[8])
{
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(PENDING));
}
template <int PENDING> __device__ __forceinline__ void fs_mx_wait(v4i_t (&r)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(PENDING));
}
template <int T> __device__ __forceinline__ v4i_t fs_mx_reduce(const v4i_t (&r)[T], const v4i_t &spa, int spidx, int bias = 0)
{
    v4i_t acc = {bias, bias, bias, bias};
#pragma unroll
    for (int t = 0; t < T; t += 2) {
        const v8i_t b = __builtin_shufflevector(r[t], r[t + 1], 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
    }
    return acc;
}

// ---- M = 64: 8-byte rows (8 queries per tile), four rows per matrix instruction ----
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) v2i_t *fs_lds_row8_t;
// one-hot operand for 8-byte rows: output row i < 8 selects bytes i, 8 + i, 16 + i, 24 + i of every lane's 32 (stored bytes
// 2 (i / 4), 4 + 2 (i / 4), 8 + 2 (i / 4), 12 + 2 (i / 4), index i % 4); rows 8..15 stay empty (their sums are 0, their thresholds 0)
__device__ __forceinline__ void fs_mx_pattern8(int i, v4i_t &a, int &idx)
{
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    uint32_t x = 0u;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        const int base = 2 * ((i & 7) >> 2);
        const bool one = (i < 8) && ((f & 3) == base) && true, pair = (i < 8) && (((f ^ 1) & 3) == base);
        if (one) w[f >> 2] |= 1u << (8 * (f & 3));
        const uint32_t v = one ? (uint32_t) (i & 3) : pair ? (uint32_t) ((i + 2) & 3) : (uint32_t) (f & 1);
        x |= v << (2 * f);
    }
    a = v4i_t{(int) w[0], (int) w[1], (int) w[2], (int) w[3]};
    idx = (int) x;
}
// address parts: LDS address of the row of lookup t = (half << 16) | (ks << 8) | (slot << 3); K[j] = bytes {slot << 3 of lookups
// 3j, 3j+1, 3j+2, half}: one v_perm_b32(K[t / 3], w, sel) builds [slot byte, ks, half, 0]
__device__ __forceinline__ void fs_mx_consts8(int lane, uint32_t (&K)[6])
{
    const int g = lane >> 4, col = lane & 15;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        uint32_t v = (uint32_t) (g >> 1) << 24;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = 3 * j + i;
            if (t < 16) v |= (uint32_t) ((((col + 16 * (g & 1) + t) & 31) << 3) & 0xff) << (8 * i);
        }
        K[j] = v;
    }
}
__device__ __forceinline__ constexpr uint32_t fs_mx_sel8(int t) { return 0x0c070000u | ((uint32_t) (t & 3) << 8) | (uint32_t) (4 + t % 3); }
template <int U> __device__ __forceinline__ uint32_t fs_mx_dword4(const u32x4 &w) { return U == 0 ? w.x : U == 1 ? w.y : U == 2 ? w.z : w.w; }
template <int OFF> __device__ __forceinline__ void fs_mx_load(u32x4 &q, const u32x4 *p) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(q) : "v"(p), "n"(OFF)); }
// compiler-scheduled form (warm-up and tail groups)
__device__ __forceinline__ v4i_t fs_mx_group8_slow(const u32x4 &w, const uint32_t (&K)[6], const v4i_t &spa, int spidx)
{
    const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
    v4i_t acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        v2i_t r[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = 4 * u + i;
            const uint32_t addr = __builtin_amdgcn_perm(K[t / 3], wd[u], fs_mx_sel8(t));
            r[i] = *(fs_lds_row8_t) (uintptr_t) addr;
        }
        const v4i_t lo = __builtin_shufflevector(r[0], r[1], 0, 1, 2, 3), hi = __builtin_shufflevector(r[2], r[3], 0, 1, 2, 3);
        const v8i_t b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
    }
    return acc;
}
// the four rows of matrix instruction U (lookups 4U .. 4U+3 = the four bytes of dword w), from asm.  (The allocator keeps two of
// the four rows outside the instruction's 8-register operand and copies them in after the wait: 4 moves per instruction.
// Pinning the rows to fixed registers removes the moves but makes it shuffle rows that are still IN FLIGHT around the
// candidate branch -- the loads are invisible to it -- so that was not kept.)
template <int U> __device__ __forceinline__ void fs_mx_issue8(uint32_t w, const uint32_t (&K)[6], v2i_t &r0, v2i_t &r1, v2i_t &r2, v2i_t &r3)
{
    uint32_t a0, a1, a2, a3;
    constexpr int t = 4 * U;
    asm volatile("v_perm_b32 %4, %8, %12, %13\n\t"
                 "v_perm_b32 %5, %9, %12, %14\n\t"
                 "v_perm_b32 %6, %10, %12, %15\n\t"
                 "v_perm_b32 %7, %11, %12, %16\n\t"
                 "ds_read_b64 %0, %4\n\t"
                 "ds_read_b64 %1, %5\n\t"
                 "ds_read_b64 %2, %6\n\t"
                 "ds_read_b64 %3, %7"
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
                 : "v"(K[t / 3]), "v"(K[(t + 1) / 3]), "v"(K[(t + 2) / 3]), "v"(K[(t + 3) / 3]), "v"(w),
                   "s"(fs_mx_sel8(t)), "s"(fs_mx_sel8(t + 1)), "s"(fs_mx_sel8(t + 2)), "s"(fs_mx_sel8(t + 3)));
}
template <int PENDING> __device__ __forceinline__ void fs_mx_wait8(v2i_t &r0, v2i_t &r1, v2i_t &r2, v2i_t &r3)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(PENDING));
}
// one matrix instruction of the rolling pipeline: wait for its four rows (PENDING younger reads stay outstanding), run it,
// refill the four registers with the same lookups' rows of the NEXT group (dword U of wn)
template <int U, int PENDING> __device__ __forceinline__ void fs_mx_step8(v2i_t (&r)[16], v4i_t &acc, const u32x4 &wn, const uint32_t (&K)[6],
                                                                         const v4i_t &spa, int spidx)
{
    fs_mx_wait8<PENDING>(r[4 * U], r[4 * U + 1], r[4 * U + 2], r[4 * U + 3]);
    const v8i_t b = {r[4 * U][0], r[4 * U][1], r[4 * U + 1][0], r[4 * U + 1][1], r[4 * U + 2][0], r[4 * U + 2][1], r[4 * U + 3][0], r[4 * U + 3][1]};
    acc = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, acc, spidx, 0, 0);
    fs_mx_issue8<U>(fs_mx_dword4<U>(wn), K, r[4 * U], r[4 * U + 1], r[4 * U + 2], r[4 * U + 3]);
}
template <int PENDING> __device__ __forceinline__ v4i_t fs_mx_group8(v2i_t (&r)[16], const u32x4 &wn, const uint32_t (&K)[6], const v4i_t &spa, int spidx,
                                                                     const v4i_t &zero)
{
    v4i_t acc = fs_acc_init(zero);
    fs_mx_step8<0, PENDING>(r, acc, wn, K, spa, spidx);
    fs_mx_step8<1, PENDING>(r, acc, wn, K, spa, spidx);
    fs_mx_step8<2, PENDING>(r, acc, wn, K, spa, spidx);
    fs_mx_step8<3, PENDING>(r, acc, wn, K, spa, spidx);
    return acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the top-1 re-rank as the TAIL of the scan (fs_tail_rerank), run by the last chunk-block of a tile.
//
// Every chunk-block of a tile appends its candidates to the tile's global candidate lists.  With TAIL the records are stored
// write-through (agent-scope stores: `sc1`), the block drains its stores (s_waitcnt vmcnt(0)) and THEN counts itself on the
// tile's arrival counter with a device-scope atomic; the block that finds chunks - 1 earlier arrivals is the last one and reads
// all records back with agent-scope (`sc1`) loads -- the write-through / `sc1`-load pairing needs no L2 write-back and no
// cache invalidate (MI355X guide, "inter-workgroup visibility": valid form {sc1 payload -> vmcnt(0) -> flag, sc1 loads}).  It
// then does exactly what rerank_top1_direct_kernel does for one query, for the <= 32 queries of the tile at once:
//   pass A  all records of the tile, flat over (query, record): the minimum quantised sum per query; records parked in LDS
//   pass B  records within the proven slack of that minimum are compacted into one survivor list (LDS)
//   pass C  one thread per survivor: exact distance from the codebook in the reference's m order (16 codeword loads in flight),
//           (orderable distance << 32 | position) minimum per query by LDS atomics
// and writes the tile's rows.  A query whose candidate buffer overflowed is scanned exhaustively by the whole block with its
// exact table in LDS (rare: thousands of duplicated nearest codes).  LDS: the byte tables are dead by then.
// One launch less per batch, no second kernel's start-up latency (8 us at B = 1024), and the rows can go straight to the host.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFsTailRec = 8192, kFsTailSurv = 4096;
constexpr size_t kFsTailLds = 32768 + 1024 + (size_t) kFsTailRec * 8 + (size_t) kFsTailSurv * 8;
// P: pointer to the kernel's own FsArgs in the kernarg segment (constant address space), laundered by the caller so that the
// tail's fields are fetched (s_load) here, behind the hot loop, instead of living in SGPRs through it
typedef const __attribute__((address_space(4))) FsArgs *fs_kernarg_t;
template <typename Vec>
__device__ __forceinline__ void fs_tail_score(fs_kernarg_t pa, const Vec *s_q, unsigned long long *s_best, int q, uint32_t n)
{
    const FsArgs &p = *(const FsArgs *) pa;
    const Vec *cw = reinterpret_cast<const Vec *>(p.tail.codewords);
    const int M = p.M;
    const uint2 *code = reinterpret_cast<const uint2 *>(p.tail.codes + (size_t) (p.tail.indirect ? (int64_t) p.tail.remap[n] : (int64_t) n) * M);
    const Vec *sq = s_q + q * M;
    float dist = 0.f;
    for (int m0 = 0; m0 < M; m0 += 16) {
        const uint2 wa = code[m0 >> 3];
        const uint2 wb = (m0 + 8 < M) ? code[(m0 >> 3) + 1] : make_uint2(0u, 0u);
        const uint32_t wd[4] = {wa.x, wa.y, wb.x, wb.y};
        Vec cv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t ks = (wd[j >> 2] >> (8 * (j & 3))) & 0xffu;
            cv[j] = cw[(m0 + j < M ? m0 + j : 0) * 256 + ks];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (m0 + j < M) dist = __fadd_rn(dist, fvec_l2sqr_vec(sq[m0 + j], cv[j]));
    }
    atomicMin(&s_best[q], ((unsigned long long) f32_orderable(__float_as_uint(dist)) << 32) | n);
}
template <typename Vec, int NQ>
__device__ __forceinline__ void fs_tail_rerank(fs_kernarg_t pa, int qbase, unsigned char *smem, int tid)
{
    const FsArgs &p = *(const FsArgs *) pa;
    constexpr int Ds = (int) (sizeof(Vec) / sizeof(float));
    const int M = p.M;
    Vec *s_q = reinterpret_cast<Vec *>(smem);                                                    // [NQ][M]: <= 32 KiB
    unsigned long long *s_best = reinterpret_cast<unsigned long long *>(smem + 32768);           // [32]
    uint32_t *s_pre = reinterpret_cast<uint32_t *>(smem + 32768 + 256);                           // [33] prefix of the usable record counts
    uint32_t *s_raw = s_pre + 40;                                                                 // [32] counts as the scan left them
    uint32_t *s_amin = s_raw + 32;                                                                // [32]
    uint32_t *s_ctl = s_amin + 32;                                                                // [0] survivors
    unsigned long long *s_rec = reinterpret_cast<unsigned long long *>(smem + 32768 + 1024);     // [kFsTailRec]
    unsigned long long *s_surv = s_rec + kFsTailRec;                                              // [kFsTailSurv] (query << 32 | position)
    const int nlive = (p.B - qbase) < NQ ? (p.B - qbase) : NQ;
    {
        const Vec *src = reinterpret_cast<const Vec *>(p.tail.queries + (int64_t) qbase * (M * Ds));
        for (int i = tid; i < nlive * M; i += kFsThreads) s_q[i] = src[i];
    }
    if (tid < 64) {                       // wave 0: counts -> exclusive prefix (lanes 0 .. 31)
        uint32_t raw = 0u;
        if (tid < nlive) raw = __hip_atomic_load(&p.cand_count[qbase + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t use = raw <= (uint32_t) p.cap ? raw : 0u;      // an overflowed list is not used at all
        uint32_t inc = use;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if ((tid & 63) >= off) inc += o;
        }
        if (tid < 32) { s_raw[tid] = raw; s_pre[tid + 1] = inc; s_best[tid] = ~0ull; s_amin[tid] = 0xffffffffu; }
        if (tid == 0) { s_pre[0] = 0u; s_ctl[0] = 0u; }
    }
    __syncthreads();
    const uint32_t T = s_pre[NQ < 32 ? NQ : 32];
    auto owner = [&](uint32_t f) {        // query of flat record f: the last q with s_pre[q] <= f
        int q = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1)
            if (q + step < NQ && s_pre[q + step] <= f) q += step;
        return q;
    };
    auto fetch = [&](int q, uint32_t f) {
        return __hip_atomic_load(&p.cand[(size_t) (qbase + q) * p.cap + (f - s_pre[q])], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (uint32_t f = tid; f < T; f += kFsThreads) {
        const int q = owner(f);
        const unsigned long long rec = fetch(q, f);
        if (f < (uint32_t) kFsTailRec) s_rec[f] = rec;
        atomicMin(&s_amin[q], (uint32_t) (rec >> 32));
    }
    __syncthreads();
    for (uint32_t f = tid; f < T; f += kFsThreads) {
        const int q = owner(f);
        const unsigned long long rec = f < (uint32_t) kFsTailRec ? s_rec[f] : fetch(q, f);
        if ((uint32_t) (rec >> 32) > s_amin[q] + (uint32_t) p.slack[qbase + q]) continue;
        const uint32_t pos = atomicAdd(&s_ctl[0], 1u);
        if (pos < (uint32_t) kFsTailSurv) s_surv[pos] = ((unsigned long long) q << 32) | (rec & 0xffffffffull);
        else fs_tail_score<Vec>(pa, s_q, s_best, q, (uint32_t) (rec & 0xffffffffull));       // (list full: scored on the spot)
    }
    __syncthreads();
    {
        const uint32_t ns = s_ctl[0] < (uint32_t) kFsTailSurv ? s_ctl[0] : (uint32_t) kFsTailSurv;
        for (uint32_t i = tid; i < ns; i += kFsThreads) {
            const unsigned long long e = s_surv[i];
            fs_tail_score<Vec>(pa, s_q, s_best, (int) (e >> 32), (uint32_t) (e & 0xffffffffull));
        }
    }
    // candidate buffer overflowed: exact table of the query in LDS (over the parked records: dead), every code scanned
    for (int q = 0; q < nlive; ++q) {
        if (s_raw[q] <= (uint32_t) p.cap) continue;              // (block-uniform)
        __syncthreads();
        float *s_tab = reinterpret_cast<float *>(s_rec);
        const Vec *cw = reinterpret_cast<const Vec *>(p.tail.codewords);
        for (int i = tid; i < M * 256; i += kFsThreads) s_tab[i] = fvec_l2sqr_vec(s_q[q * M + (i >> 8)], cw[i]);
        __syncthreads();
        unsigned long long best = ~0ull;
        for (int64_t n = tid; n < p.n_codes; n += kFsThreads) {
            const float d = exact_adist(s_tab, p.tail.codes + (size_t) (p.tail.indirect ? (int64_t) p.tail.remap[n] : (int64_t) n) * M, M, 256);
            const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(d)) << 32) | (uint32_t) n;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = o < best ? o : best;
        }
        if ((tid & 63) == 0 && best != ~0ull) atomicMin(&s_best[q], best);
    }
    __syncthreads();
    if (tid < nlive) {
        const unsigned long long k = s_best[tid];
        const uint32_t idx = (uint32_t) (k & 0xffffffffu);
        const int64_t b = qbase + tid;
        p.tail.out_ids[b * p.tail.topk] = (k == ~0ull) ? -1 : (p.tail.remap ? p.tail.remap[idx] : (int64_t) idx);
        p.tail.out_dists[b * p.tail.topk] = (k == ~0ull) ? INFINITY : __uint_as_float(f32_unorderable((uint32_t) (k >> 32)));
    }
    if (p.tail.host_flag) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the rows (coherent host memory) have left the CU ahead of the flag
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();
            __hip_atomic_store(&p.tail.host_flag[blockIdx.y], p.tail.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// arrival of a chunk-block at its tile's counter; true for the last one (which also puts the zero back for the next launch).
// s_word: one LDS word for the broadcast.
// spilled: some record of this block went to global memory with a plain store (LDS staging area full): one L2 write-back first.
__device__ __forceinline__ bool fs_tail_arrive(fs_kernarg_t pa, uint32_t *s_word, int tid, bool spilled)
{
    const FsArgs &p = *(const FsArgs *) pa;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this thread's candidate stores have left the CU
    __syncthreads();
    if (tid == 0) {
        if (spilled) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler may drop the wait behind buffer_wbl2: MI355X guide)
        }
        const unsigned int prev = __hip_atomic_fetch_add(&p.tail.tile_done[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) __hip_atomic_store(&p.tail.tile_done[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_word = prev;
    }
    __syncthreads();
    return *s_word == gridDim.x - 1;
}

constexpr int kFsMxSeg = 256;        // MODE 1 segments per chunk and query: (wave, column) pairs

// grid = (chunks, ceil(B / 16)), 1024 threads.  Thresholds: 16 words in LDS, candidate <=> a < thr (thr = bound + slack + 1).
template <int T, int MODE, int QR = 16, bool TAIL = false, bool PIPE = false>
__global__ __launch_bounds__(kFsThreads) void fscan_mx_kernel(FsArgs p)
{
    static_assert(!TAIL || (MODE == 0 && QR == 16), "the fused re-rank is the top-1 pass of the 16-byte-row shapes");
    static_assert(!PIPE || (MODE == 0 && QR == 16), "the one-group-late judge is the top-1 pass of the 16-byte-row shapes (unsigned table bytes)");
    static_assert((QR == 16 && (T == 4 || T == 8)) || (QR == 8 && T == 16), "M = 16 / 32 with 16-byte rows, M = 64 with 8-byte rows");
    constexpr int M = 4 * T;
    typedef typename FsMxW<T>::V W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, tile = blockIdx.y;
    constexpr size_t lut_bytes = (size_t) M * 256 * QR;
    uint32_t *s_thr = reinterpret_cast<uint32_t *>(smem + lut_bytes);                       // [16]
    uint32_t *s_lcnt = reinterpret_cast<uint32_t *>(smem + lut_bytes + 64);                  // [16] staged, [16] global bases
    uint32_t *s_slk = reinterpret_cast<uint32_t *>(smem + lut_bytes + 64 + QR * 8);          // [16] the queries' slacks (the candidate path
                                                                                             //      read them from global memory: ~1 us each)
    unsigned long long *s_lcand = reinterpret_cast<unsigned long long *>(smem + lut_bytes + 64 + QR * 8 + 64);
    {
        if (tid < 16) s_slk[tid] = (tid < QR && tile * QR + tid < p.B) ? (uint32_t) p.slack[tile * QR + tid] : 0u;
        if (QR == 16 && p.quarter) {
            // four quarter tables (one dword = the levels of four queries for one (m, ks)) -> 16-byte rotated rows.  Lane =
            // (slot = m mod 16, quarter): the 64 lanes of a wave write the 64 dwords of 16 consecutive rows (one ks, one half) --
            // 256 contiguous LDS bytes per ds_write_b32, conflict-free; each lane reads 64 contiguous bytes (16 ks) of its
            // quarter table per unit.
            const uint32_t *src = reinterpret_cast<const uint32_t *>(p.qlut) + (size_t) tile * 4 * M * 256;
            const int slot = tid & 15, qq = (tid >> 4) & 3;
            for (int unit = tid >> 6; unit < M; unit += kFsThreads >> 6) {          // unit = (half, block of 16 ks): M / 16 x 16 of them
                const int h = unit >> 4, ks0 = (unit & 15) * 16;
                const uint4 *sp = reinterpret_cast<const uint4 *>(src + ((size_t) qq * M + 16 * h + slot) * 256 + ks0);
                const uint4 v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3];
                uint32_t *d = reinterpret_cast<uint32_t *>(smem + ((size_t) h * 4096 + (size_t) ks0 * 16 + slot) * 16 + qq * 4);
                d[0 * 64] = v0.x; d[1 * 64] = v0.y; d[2 * 64] = v0.z; d[3 * 64] = v0.w;
                d[4 * 64] = v1.x; d[5 * 64] = v1.y; d[6 * 64] = v1.z; d[7 * 64] = v1.w;
                d[8 * 64] = v2.x; d[9 * 64] = v2.y; d[10 * 64] = v2.z; d[11 * 64] = v2.w;
                d[12 * 64] = v3.x; d[13 * 64] = v3.y; d[14 * 64] = v3.z; d[15 * 64] = v3.w;
            }
        } else {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(p.qlut + (size_t) tile * lut_bytes);
            uint4 *d4 = reinterpret_cast<uint4 *>(smem);
            for (size_t i = tid; i < lut_bytes / 16; i += kFsThreads) d4[i] = s4[i];
        }
        if (tid < 16) {
            // queries past the end of the batch (last tile) and the unused rows of an 8-query tile get threshold 0: they never hit
            const int b = tile * QR + tid;
            const bool live = tid < QR && b < p.B;
            uint32_t t = live ? 0xffffu : 0u;
            if constexpr (MODE == 2) t = live ? p.thr16[b] : 0u;
            s_thr[tid] = t;
        }
        if (tid < 2 * QR) s_lcnt[tid] = 0u;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, col = lane & 15, gq = lane >> 4;      // this lane judges queries 4 gq .. 4 gq + 3
    v4i_t spa;
    int spidx;
    uint32_t C[QR == 16 ? T : 6];
    if constexpr (QR == 16) {
        fs_mx_pattern(col, spa, spidx);
        fs_mx_consts<T>(lane, C);
    } else {
        fs_mx_pattern8(col, spa, spidx);
        fs_mx_consts8(lane, C);
    }

    const int64_t c_begin = (int64_t) blockIdx.x * p.chunk_len;         // a multiple of 1024
    int64_t c_end = c_begin + p.chunk_len;
    if (c_end > p.n_codes) c_end = p.n_codes;
    const int64_t span = c_end > c_begin ? c_end - c_begin : 0;
    const int full = (int) (span / kFsThreads);                         // trips of 64 whole groups (1024 codes)
    const int tail_groups = (int) ((span - (int64_t) full * kFsThreads + 15) / 16);
    const W *fc = reinterpret_cast<const W *>(p.codes) + (size_t) (c_begin / 16) * 64 + lane;

    auto emit = [&](int q, uint32_t a, uint32_t t, uint32_t n) {
        const int b = tile * QR + q;
        if (b >= p.B) return;
        if constexpr (MODE == 0) {
            const uint32_t nt = fs_thr_of(a, s_slk[q]);
            if (nt < t) {
                atomicMin(&s_thr[q], nt);
                atomicMin(&p.gthr[b], nt);       // let the other chunks of this tile prune with it too
            }
        }
        const unsigned long long rec = ((unsigned long long) a << 32) | n;
        bool staged = false;
        if (p.lcap > 0) {
            const unsigned int lp = atomicAdd(&s_lcnt[q], 1u);
            if (lp < (unsigned int) p.lcap) { s_lcand[(size_t) q * p.lcap + lp] = rec; staged = true; }
        }
        if (!staged) {
            const unsigned int pos = atomicAdd(&p.cand_count[b], 1u);
            if (pos < (unsigned int) p.cap) p.cand[(size_t) b * p.cap + pos] = rec;
        }
    };
    // the sums of one group (code n = first + col) against the thresholds
    auto judge = [&](const v4i_t &acc, const v4i_t &thr, uint32_t n) {
        const bool h0 = acc[0] < thr[0], h1 = acc[1] < thr[1], h2 = acc[2] < thr[2], h3 = acc[3] < thr[3];
        if (h0 | h1 | h2 | h3) {
            // rare path: walk the queries that hit (usually exactly one), one emission site
            uint32_t mask = (h0 ? 1u : 0u) | (h1 ? 2u : 0u) | (h2 ? 4u : 0u) | (h3 ? 8u : 0u);
            while (mask) {
                const int r = __ffs((int) mask) - 1;
                mask &= mask - 1u;
                const int a = r == 0 ? acc[0] : r == 1 ? acc[1] : r == 2 ? acc[2] : acc[3];
                const int t = r == 0 ? thr[0] : r == 1 ? thr[1] : r == 2 ? thr[2] : thr[3];
                emit(4 * gq + r, (uint32_t) a, (uint32_t) t, n);
            }
        }
    };
    v4i_t keep = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};     // MODE 0: warm-up minima; MODE 1: segment minima
    auto take_min = [&](const v4i_t &acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[r] = acc[r] < keep[r] ? acc[r] : keep[r];
    };
    // MODE 0 warm-up: the minima of the block's first trip become the first thresholds
    auto publish = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int v = keep[r];
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                const int o = __shfl_xor(v, off);
                v = o < v ? o : v;
            }
            const int b = tile * QR + 4 * gq + r;
            if (col == 0 && 4 * gq + r < QR && v != 0x7fffffff && b < p.B)
                atomicMin(&s_thr[4 * gq + r], fs_thr_of((uint32_t) v, s_slk[4 * gq + r]));
        }
    };
    auto load_thr = [&]() { return *reinterpret_cast<const v4i_t *>(s_thr + 4 * gq); };
    // adopt thresholds published by the blocks scanning the other chunks for the same queries (stale = a few more candidates)
    // (wave 0 does it; taking turns over the 16 waves was measured in round 3: no gain at B = 1024, 4 % slower at B = 128)
    auto adopt = [&](bool first) {
        const int q = tid & 63;
        if (MODE == 0 && tid < 64 && q < QR) {
            const int b = tile * QR + q;
            if (b < p.B) {
                const uint32_t g = __hip_atomic_load(&p.gthr[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (first) {
                    const uint32_t mine = s_thr[q];
                    if (mine < g) atomicMin(&p.gthr[b], mine);
                }
                atomicMin(&s_thr[q], g);
            }
        }
    };
    // one group at a time (warm-up, tail): group index gi counted from the chunk's first group
    auto slow_group = [&](int64_t gi, bool minima, const v4i_t &thr) {
        const W w = fc[(size_t) gi * 64];
        v4i_t acc;
        if constexpr (QR == 16) {
            v4i_t r[T];
            fs_mx_issue<T>(w, C, r);
            acc = fs_mx_reduce<T>(r, spa, spidx, p.bias);
        } else {
            acc = fs_mx_group8_slow(w, C, spa, spidx);
        }
        const int64_t n = c_begin + gi * 16 + col;
        if (n >= c_end) acc = v4i_t{0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};     // columns past the end are never judged
        if (minima) take_min(acc);
        else judge(acc, thr, (uint32_t) n);
    };
    auto tail = [&](bool minima) {
        const v4i_t thr = load_thr();
        for (int gi = wave; gi < tail_groups; gi += kFsThreads / 64) slow_group((int64_t) full * 64 + gi, minima, thr);
    };

    // MODE 0 warm-up: the minima of the chunk's first trip (or of the tail, when that is all there is) become the first
    // thresholds before anything is judged; that trip is then judged LAST, with thresholds that know the whole chunk
    if constexpr (MODE == 0) {
        if (full > 0) {
            const v4i_t none = {0, 0, 0, 0};
#pragma unroll 1
            for (int j = 0; j < 4; ++j) slow_group(wave * 4 + j, true, none);      // (the wave's four groups of the first trip; fewer: more candidates, DESIGN 3.1 v)
        } else {
            tail(true);
        }
        publish();
        __syncthreads();
        adopt(true);
        __syncthreads();
    }
    // whole trips, software-pipelined: a wave takes four consecutive groups per trip; the rows of group j + 1 are in flight
    // while the matrix instructions of group j run, the lookups of the next trip's groups while this trip's are consumed.
    // Order: MODE 0: 1, 2, .., full - 1, 0;  MODE 2: 0 .. full - 1;  MODE 1: every sample_stride-th (an upper bound needs
    // only a sample)
    const int step = (MODE == 1) ? p.sample_stride : 1;
    const int ntrip = (MODE == 1) ? (full + step - 1) / step : full;
    auto trip_of = [&](int k) { return (MODE == 0) ? (k + 1 < full ? k + 1 : 0) : k * step; };
    if (ntrip > 0) {
        // a zero accumulator kept in registers: the matrix instruction accumulates in place, and built from a literal the
        // compiler clears it with six moves per group instead of two
        v4i_t zero4 = {p.bias, p.bias, p.bias, p.bias};       // (128 M when the table bytes are signed: see qlut_fused_kernel)
        asm volatile("" : "+v"(zero4));
        // Per wave the groups form one sequence g = 4 * trip + j.  Two register sets of T rows alternate (even / odd groups);
        // while group g runs through the matrix core, the rows of g + 1 are in flight and those of g + 2 are being issued into
        // g's own registers (fs_mx_reduce_refill).  q[j] holds the lookups of the next group of column j that still needs
        // its rows fetched; it is reloaded right after use, a whole trip ahead of its next use.
        const W *pw = fc + (size_t) wave * 4 * 64;
        auto trip_ptr = [&](int k) { return pw + (size_t) trip_of(k < ntrip ? k : ntrip - 1) * 64 * 64; };   // past the end: a valid address, rows never used
        if constexpr (QR == 16) {
            v4i_t ra[T], rb[T];
            W q[4];
            {
                const W *p0 = trip_ptr(0), *p1 = trip_ptr(1);
                const W g0 = p0[0], g1 = p0[64];
                fs_mx_issue_hot<T>(g0, C, ra);
                fs_mx_issue_hot<T>(g1, C, rb);
                constexpr int S = 64 * (int) sizeof(W);
                fs_mx_load<2 * S>(q[2], p0);        // in the order of their use: every use has three younger loads behind it
                fs_mx_load<3 * S>(q[3], p0);
                fs_mx_load<0>(q[0], p1);
                fs_mx_load<S>(q[1], p1);
            }
            const uint32_t thr_addr = (uint32_t) (lut_bytes + 16 * gq);       // s_thr + 4 * gq as an LDS address (dynamic LDS starts at 0)
            v4i_t thr;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(thr) : "v"(thr_addr));     // (not a compiler-visible load: it would be waited for inside the loop)
            // Round 4: the sums of a group are judged ONE GROUP LATER, behind the matrix instructions of the next group -- a vector
            // compare right behind the last v_smfmac of its own group waits out the matrix pipe's result latency (the compiler pads it
            // with `s_nop 7`: ~11 idle issue cycles per group and wave, 6 % of a wave's time per group).  Two accumulator sets alternate;
            // the registers come from the zero accumulator, which is a literal when the table bytes are unsigned (bias == 0: 63 / 127
            // levels); signed tables (255 levels: bias = 128 M) keep the in-order judge.  Any threshold read later than before is still
            // an upper bound, so the candidates stay a superset of what the proof needs.
            if constexpr (PIPE) {
                auto hot = [&](auto zlit) {
                    constexpr bool ZLIT = decltype(zlit)::value;
                    v4i_t zero_l = {0, 0, 0, 0};
                    const v4i_t &zz = ZLIT ? zero_l : zero4;
                    v4i_t accp = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};      // (nothing pending: never below a threshold)
                    uint32_t np = 0u;
                    for (int k = 0; k < ntrip; ++k) {
                        const int it = trip_of(k);
                        const W *pn = trip_ptr(k + 1), *pnn = trip_ptr(k + 2);
                        const uint32_t n = (uint32_t) c_begin + (uint32_t) it * kFsThreads + wave * 64 + col;     // positions fit 32 bits (the records hold 32)
                        constexpr int S = 64 * (int) sizeof(W);                // bytes between the lookups of consecutive groups
                        auto settle = [&](const v4i_t &a, uint32_t nn) { if (MODE == 1) take_min(a); else judge(a, thr, nn); };
                        fs_mx_vmwait<3>(q[2]);
                        fs_mx_wait<T>(ra);                                    // group 0 (the T younger reads are group 1's)
                        const v4i_t acc0 = fs_mx_reduce_refill<T, ZLIT>(ra, q[2], C, spa, spidx, zz);   // ... refilled with group 2's rows
                        fs_mx_load<2 * S>(q[2], pn);
                        if constexpr (ZLIT) settle(accp, np); else settle(acc0, n);
                        fs_mx_vmwait<3>(q[3]);
                        fs_mx_wait<T>(rb);                                    // group 1
                        const v4i_t acc1 = fs_mx_reduce_refill<T, ZLIT>(rb, q[3], C, spa, spidx, zz);
                        fs_mx_load<3 * S>(q[3], pn);
                        if constexpr (ZLIT) settle(acc0, n); else settle(acc1, n + 16);
                        // thresholds: re-read once per trip, straight into the registers group 2 may still be comparing against -- any mix
                        // of old and new words is a valid set of thresholds (each word is an upper bound at all times)
                        if constexpr (MODE == 0) asm volatile("ds_read_b128 %0, %1 ; rii:inflight-ok (tools/check_isa_inflight.py)" : "+v"(thr) : "v"(thr_addr));
                        fs_mx_vmwait<3>(q[0]);
                        fs_mx_wait<(MODE == 0) ? T + 1 : T>(ra);              // group 2 (younger: group 3's rows and the thresholds)
                        const v4i_t acc2 = fs_mx_reduce_refill<T, ZLIT>(ra, q[0], C, spa, spidx, zz);   // next trip's group 0
                        fs_mx_load<0>(q[0], pnn);
                        if constexpr (ZLIT) settle(acc1, n + 16); else settle(acc2, n + 32);
                        fs_mx_vmwait<3>(q[1]);
                        fs_mx_wait<T>(rb);                                    // group 3 (and the thresholds: older than group 2's refills)
                        const v4i_t acc3 = fs_mx_reduce_refill<T, ZLIT>(rb, q[1], C, spa, spidx, zz);
                        fs_mx_load<S>(q[1], pnn);
                        if constexpr (ZLIT) { settle(acc2, n + 32); accp = acc3; np = n + 48; } else settle(acc3, n + 48);
                        adopt(false);
                    }
                    if constexpr (ZLIT) { if (MODE == 1) take_min(accp); else judge(accp, thr, np); }
                };
                hot(std::true_type{});
            } else {
                for (int k = 0; k < ntrip; ++k) {
                    const int it = trip_of(k);
                    const W *pn = trip_ptr(k + 1), *pnn = trip_ptr(k + 2);
                    const uint32_t n = (uint32_t) c_begin + (uint32_t) it * kFsThreads + wave * 64 + col;     // positions fit 32 bits (the records hold 32)
                    v4i_t acc;
                    constexpr int S = 64 * (int) sizeof(W);                // bytes between the lookups of consecutive groups
                    fs_mx_vmwait<3>(q[2]);
                    fs_mx_wait<T>(ra);                                    // group 0 (the T younger reads are group 1's)
                    acc = fs_mx_reduce_refill<T>(ra, q[2], C, spa, spidx, zero4);   // ... refilled with group 2's rows
                    fs_mx_load<2 * S>(q[2], pn);
                    if (MODE == 1) take_min(acc); else judge(acc, thr, n);
                    fs_mx_vmwait<3>(q[3]);
                    fs_mx_wait<T>(rb);                                    // group 1
                    acc = fs_mx_reduce_refill<T>(rb, q[3], C, spa, spidx, zero4);
                    fs_mx_load<3 * S>(q[3], pn);
                    if (MODE == 1) take_min(acc); else judge(acc, thr, n + 16);
                    // thresholds: re-read once per trip, straight into the registers group 2 may still be comparing against -- any mix
                    // of old and new words is a valid set of thresholds (each word is an upper bound at all times)
                    if constexpr (MODE == 0) asm volatile("ds_read_b128 %0, %1 ; rii:inflight-ok (tools/check_isa_inflight.py)" : "+v"(thr) : "v"(thr_addr));
                    fs_mx_vmwait<3>(q[0]);
                    fs_mx_wait<(MODE == 0) ? T + 1 : T>(ra);              // group 2 (younger: group 3's rows and the thresholds)
                    acc = fs_mx_reduce_refill<T>(ra, q[0], C, spa, spidx, zero4);   // next trip's group 0
                    fs_mx_load<0>(q[0], pnn);
                    if (MODE == 1) take_min(acc); else judge(acc, thr, n + 32);
                    fs_mx_vmwait<3>(q[1]);
                    fs_mx_wait<T>(rb);                                    // group 3 (and the thresholds: older than group 2's refills)
                    acc = fs_mx_reduce_refill<T>(rb, q[1], C, spa, spidx, zero4);
                    fs_mx_load<S>(q[1], pnn);
                    if (MODE == 1) take_min(acc); else judge(acc, thr, n + 48);
                    adopt(false);
                }
            }
            fs_mx_wait<0>(ra);            // the rows and lookups fetched past the last trip are never used, but must have landed
            fs_mx_wait<0>(rb);
            fs_mx_vmwait<0>(q[0]); fs_mx_vmwait<0>(q[1]); fs_mx_vmwait<0>(q[2]); fs_mx_vmwait<0>(q[3]);
    
        } else {
            // M = 64: one register set of 16 rows, rolling at matrix-instruction granularity (the LDS counter of a wave holds 15
            // operations, so two groups of 16 reads cannot be in flight anyway): instruction u of group j waits for its four
            // rows -- the twelve reads issued since stay outstanding -- runs, and its registers take instruction u's rows of
            // group j + 1.  q[j]: lookups of column j's next group, reloaded a trip ahead as above.
            v2i_t r[16];
            W q[4];
            constexpr int S = 64 * (int) sizeof(W);
            {
                const W *p0 = trip_ptr(0), *p1 = trip_ptr(1);
                const W g0 = p0[0];
                fs_mx_issue8<0>(g0.x, C, r[0], r[1], r[2], r[3]);
                fs_mx_issue8<1>(g0.y, C, r[4], r[5], r[6], r[7]);
                fs_mx_issue8<2>(g0.z, C, r[8], r[9], r[10], r[11]);
                fs_mx_issue8<3>(g0.w, C, r[12], r[13], r[14], r[15]);
                fs_mx_load<S>(q[1], p0);            // in the order of their use: every use has three younger loads behind it
                fs_mx_load<2 * S>(q[2], p0);
                fs_mx_load<3 * S>(q[3], p0);
                fs_mx_load<0>(q[0], p1);
            }
            const uint32_t thr_addr = (uint32_t) (lut_bytes + 16 * gq);
            v4i_t thr;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(thr) : "v"(thr_addr));
            for (int k = 0; k < ntrip; ++k) {
                const int it = trip_of(k);
                const W *pn = trip_ptr(k + 1), *pnn = trip_ptr(k + 2);
                const uint32_t n = (uint32_t) c_begin + (uint32_t) it * kFsThreads + wave * 64 + col;
                v4i_t acc;
                fs_mx_vmwait<3>(q[1]);
                acc = fs_mx_group8<12>(r, q[1], C, spa, spidx, zero4);          // group 0; refills = group 1's rows
                fs_mx_load<S>(q[1], pn);
                if (MODE == 1) take_min(acc); else judge(acc, thr, n);
                fs_mx_vmwait<3>(q[2]);
                acc = fs_mx_group8<12>(r, q[2], C, spa, spidx, zero4);          // group 1
                fs_mx_load<2 * S>(q[2], pn);
                if (MODE == 1) take_min(acc); else judge(acc, thr, n + 16);
                // thresholds: re-read once per trip into the live registers (any mix of old and new words is valid); the read
                // sits behind group 2's rows in the queue, so group 2's four waits see one more outstanding operation
                if constexpr (MODE == 0) asm volatile("ds_read_b128 %0, %1 ; rii:inflight-ok (tools/check_isa_inflight.py)" : "+v"(thr) : "v"(thr_addr));
                fs_mx_vmwait<3>(q[3]);
                acc = fs_mx_group8<(MODE == 0) ? 13 : 12>(r, q[3], C, spa, spidx, zero4);   // group 2
                fs_mx_load<3 * S>(q[3], pn);
                if (MODE == 1) take_min(acc); else judge(acc, thr, n + 32);
                fs_mx_vmwait<3>(q[0]);
                acc = fs_mx_group8<12>(r, q[0], C, spa, spidx, zero4);          // group 3; refills = next trip's group 0
                fs_mx_load<0>(q[0], pnn);
                if (MODE == 1) take_min(acc); else judge(acc, thr, n + 48);
                adopt(false);
            }
            fs_mx_wait8<0>(r[0], r[1], r[2], r[3]);       // everything fetched past the last trip has landed
            fs_mx_vmwait<0>(q[0]); fs_mx_vmwait<0>(q[1]); fs_mx_vmwait<0>(q[2]); fs_mx_vmwait<0>(q[3]);
        }
    }
    tail(MODE == 1);

    if (MODE == 0 && p.lcap > 0) {
        __syncthreads();
        if (tid < QR) {
            const int b = tile * QR + tid;
            const unsigned int c = min(s_lcnt[tid], (unsigned int) p.lcap);
            s_lcnt[QR + tid] = (c && b < p.B) ? atomicAdd(&p.cand_count[b], c) : 0u;
        }
        __syncthreads();
        for (int q = tid >> 6; q < QR; q += kFsThreads >> 6) {        // one wave per query
            const int b = tile * QR + q;
            const unsigned int c = min(s_lcnt[q], (unsigned int) p.lcap), base = s_lcnt[QR + q];
            for (unsigned int i = tid & 63; i < c; i += 64)
                if (base + i < (unsigned int) p.cap) {
                    if constexpr (TAIL) fs_cand_store(&p.cand[(size_t) b * p.cap + base + i], s_lcand[(size_t) q * p.lcap + i]);     // write-through: read by the tile's last block
                    else p.cand[(size_t) b * p.cap + base + i] = s_lcand[(size_t) q * p.lcap + i];
                }
        }
    }
    if constexpr (TAIL) {
        // the last chunk-block of the tile re-ranks the tile's 16 queries (fs_tail_rerank); everybody else is done
        fs_kernarg_t pa = (fs_kernarg_t) __builtin_amdgcn_kernarg_segment_ptr();     // FsArgs is the only parameter
        asm volatile("" : "+s"(pa));
        bool spilled = false;
        for (int q = 0; q < 16; ++q) spilled |= s_lcnt[q] > (uint32_t) pa->lcap;
        if (fs_tail_arrive(pa, s_lcnt, tid, spilled)) {
            if (pa->tail.Ds == 4) fs_tail_rerank<float4, 16>(pa, tile * 16, smem, tid);
            else fs_tail_rerank<float2, 16>(pa, tile * 16, smem, tid);
        }
    }
    if constexpr (MODE == 1) {
        const size_t G = (size_t) gridDim.x * kFsMxSeg;
        const size_t seg = (size_t) blockIdx.x * kFsMxSeg + wave * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = tile * QR + 4 * gq + r;
            if (4 * gq + r < QR && b < p.B) p.segmin[(size_t) b * G + seg] = (uint16_t) (keep[r] > 0xffff ? 0xffff : keep[r]);
        }
    }
}

// =====================================================================================================================
// fscan_mx_dual_kernel: M = 16 with TWO 16-query tiles resident (round 3).  The byte tables of 16 queries are 64 KiB at M = 16, so
// fscan_mx_kernel<4> leaves half of the LDS unused and keeps only 8 row reads in flight per wave (two register sets of 4 rows):
// 0.50 - 0.56 of the LDS row rate against 0.65 for M = 32.  Here a block holds the tables of tiles 2y and 2y + 1 (128 KiB: the
// second tile 64 KiB above the first, which is exactly the `half` bit of the row address), every lane turns each of its four
// code bytes into TWO row addresses -- the same v_perm_b32 with a second per-lane constant -- and the eight rows of a group go
// through the matrix core as two pairs of instructions with separate accumulators.  The pipeline is the M = 32 one (two register
// sets of 8 rows, 16 reads in flight); one fetch of the lookups serves 32 queries, so the shard is streamed B / 32 times
// instead of B / 16: half the HBM traffic of the Deep1B-shaped scan.  Thresholds: 32 words in LDS.
// =====================================================================================================================
__device__ __forceinline__ void fs_mx_issue_hot_dual(uint32_t w, const uint32_t (&C)[8], v4i_t (&r)[8])
{
    fs_mx_issue2<0>(w, C[0], C[1], r[0], r[1]);
    fs_mx_issue2<2>(w, C[2], C[3], r[2], r[3]);
    fs_mx_issue2<0>(w, C[4], C[5], r[4], r[5]);
    fs_mx_issue2<2>(w, C[6], C[7], r[6], r[7]);
}
// rows r[0..3] -> tile A's sums, r[4..7] -> tile B's; every pair of registers is refilled with the rows of the group two ahead
// (lookups wn) right behind the instruction that consumed it
__device__ __forceinline__ void fs_mx_reduce_refill_dual(v4i_t (&r)[8], uint32_t wn, const uint32_t (&C)[8], const v4i_t &spa, int spidx,
                                                         const v4i_t &zero, v4i_t &accA, v4i_t &accB)
{
    accA = fs_acc_init(zero);
    {
        const v8i_t b = __builtin_shufflevector(r[0], r[1], 0, 1, 2, 3, 4, 5, 6, 7);
        accA = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, accA, spidx, 0, 0);
        fs_mx_issue2<0>(wn, C[0], C[1], r[0], r[1]);
    }
    {
        const v8i_t b = __builtin_shufflevector(r[2], r[3], 0, 1, 2, 3, 4, 5, 6, 7);
        accA = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, accA, spidx, 0, 0);
        fs_mx_issue2<2>(wn, C[2], C[3], r[2], r[3]);
    }
    accB = fs_acc_init(zero);
    {
        const v8i_t b = __builtin_shufflevector(r[4], r[5], 0, 1, 2, 3, 4, 5, 6, 7);
        accB = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, accB, spidx, 0, 0);
        fs_mx_issue2<0>(wn, C[4], C[5], r[4], r[5]);
    }
    {
        const v8i_t b = __builtin_shufflevector(r[6], r[7], 0, 1, 2, 3, 4, 5, 6, 7);
        accB = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, b, accB, spidx, 0, 0);
        fs_mx_issue2<2>(wn, C[6], C[7], r[6], r[7]);
    }
}

// grid = (chunks, ceil(B / 32)), 1024 threads
template <int MODE, bool TAIL = false>
__global__ __launch_bounds__(kFsThreads) void fscan_mx_dual_kernel(FsArgs p)
{
    static_assert(!TAIL || MODE == 0, "the fused re-rank is the top-1 pass");
    constexpr int M = 16, NQ = 32;
    constexpr size_t tile_bytes = (size_t) M * 256 * 16;               // 64 KiB: one tile's rotated byte rows
    constexpr size_t lut_bytes = 2 * tile_bytes;
    typedef uint32_t W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, tile2 = blockIdx.y;                   // this block: tiles 2 tile2, 2 tile2 + 1 = queries 32 tile2 .. + 31
    const int qbase = tile2 * NQ;
    uint32_t *s_thr = reinterpret_cast<uint32_t *>(smem + lut_bytes);                       // [32]
    uint32_t *s_lcnt = reinterpret_cast<uint32_t *>(smem + lut_bytes + NQ * 4);              // [32] staged, [32] global bases
    uint32_t *s_slk = reinterpret_cast<uint32_t *>(smem + lut_bytes + NQ * 4 + NQ * 8);      // [32] the queries' slacks
    unsigned long long *s_lcand = reinterpret_cast<unsigned long long *>(smem + lut_bytes + NQ * 4 + NQ * 8 + NQ * 4);
    {
        if (tid < NQ) s_slk[tid] = (qbase + tid < p.B) ? (uint32_t) p.slack[qbase + tid] : 0u;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t tile = 2 * (int64_t) tile2 + u;
            unsigned char *dstb = smem + u * tile_bytes;
            if (tile * 16 >= p.B) {                                     // no second tile (odd tile count): all-zero rows, thresholds 0
                uint4 *d4 = reinterpret_cast<uint4 *>(dstb);
                for (size_t i = tid; i < tile_bytes / 16; i += kFsThreads) d4[i] = make_uint4(0u, 0u, 0u, 0u);
            } else if (p.quarter) {                                     // quarter tables -> rotated rows (see fscan_mx_kernel)
                const uint32_t *src = reinterpret_cast<const uint32_t *>(p.qlut) + (size_t) tile * 4 * M * 256;
                const int slot = tid & 15, qq = (tid >> 4) & 3;
                for (int unit = tid >> 6; unit < M; unit += kFsThreads >> 6) {
                    const int ks0 = unit * 16;
                    const uint4 *sp = reinterpret_cast<const uint4 *>(src + ((size_t) qq * M + slot) * 256 + ks0);
                    const uint4 v0 = sp[0], v1 = sp[1], v2 = sp[2], v3 = sp[3];
                    uint32_t *d = reinterpret_cast<uint32_t *>(dstb + ((size_t) ks0 * 16 + slot) * 16 + qq * 4);
                    d[0 * 64] = v0.x; d[1 * 64] = v0.y; d[2 * 64] = v0.z; d[3 * 64] = v0.w;
                    d[4 * 64] = v1.x; d[5 * 64] = v1.y; d[6 * 64] = v1.z; d[7 * 64] = v1.w;
                    d[8 * 64] = v2.x; d[9 * 64] = v2.y; d[10 * 64] = v2.z; d[11 * 64] = v2.w;
                    d[12 * 64] = v3.x; d[13 * 64] = v3.y; d[14 * 64] = v3.z; d[15 * 64] = v3.w;
                }
            } else {
                const uint4 *s4 = reinterpret_cast<const uint4 *>(p.qlut + (size_t) tile * tile_bytes);
                uint4 *d4 = reinterpret_cast<uint4 *>(dstb);
                for (size_t i = tid; i < tile_bytes / 16; i += kFsThreads) d4[i] = s4[i];
            }
        }
        if (tid < NQ) {
            const int b = qbase + tid;
            const bool live = b < p.B;
            uint32_t t = live ? 0xffffu : 0u;
            if constexpr (MODE == 2) t = live ? p.thr16[b] : 0u;
            s_thr[tid] = t;
        }
        if (tid < 2 * NQ) s_lcnt[tid] = 0u;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, col = lane & 15, gq = lane >> 4;      // this lane judges queries 4 gq .. + 3 of both tiles
    v4i_t spa;
    int spidx;
    uint32_t C[8];
    fs_mx_pattern(col, spa, spidx);
    {
        uint32_t C4[4];
        fs_mx_consts<4>(lane, C4);
#pragma unroll
        for (int t = 0; t < 4; ++t) { C[t] = C4[t]; C[4 + t] = C4[t] | 0x10000u; }      // tile B: 64 KiB above tile A
    }
    const int64_t c_begin = (int64_t) blockIdx.x * p.chunk_len;         // a multiple of 1024
    int64_t c_end = c_begin + p.chunk_len;
    if (c_end > p.n_codes) c_end = p.n_codes;
    const int64_t span = c_end > c_begin ? c_end - c_begin : 0;
    const int full = (int) (span / kFsThreads);
    const int tail_groups = (int) ((span - (int64_t) full * kFsThreads + 15) / 16);
    const W *fc = reinterpret_cast<const W *>(p.codes) + (size_t) (c_begin / 16) * 64 + lane;

    auto emit = [&](int q, uint32_t a, uint32_t t, uint32_t n) {
        const int b = qbase + q;
        if (b >= p.B) return;
        if constexpr (MODE == 0) {
            const uint32_t nt = fs_thr_of(a, s_slk[q]);
            if (nt < t) {
                atomicMin(&s_thr[q], nt);
                atomicMin(&p.gthr[b], nt);
            }
        }
        const unsigned long long rec = ((unsigned long long) a << 32) | n;
        bool staged = false;
        if (p.lcap > 0) {
            const unsigned int lp = atomicAdd(&s_lcnt[q], 1u);
            if (lp < (unsigned int) p.lcap) { s_lcand[(size_t) q * p.lcap + lp] = rec; staged = true; }
        }
        if (!staged) {
            const unsigned int pos = atomicAdd(&p.cand_count[b], 1u);
            if (pos < (unsigned int) p.cap) p.cand[(size_t) b * p.cap + pos] = rec;
        }
    };
    auto judge = [&](const v4i_t &acc, const v4i_t &thr, uint32_t n, int qoff) {
        const bool h0 = acc[0] < thr[0], h1 = acc[1] < thr[1], h2 = acc[2] < thr[2], h3 = acc[3] < thr[3];
        if (h0 | h1 | h2 | h3) {
            uint32_t mask = (h0 ? 1u : 0u) | (h1 ? 2u : 0u) | (h2 ? 4u : 0u) | (h3 ? 8u : 0u);
            while (mask) {
                const int r = __ffs((int) mask) - 1;
                mask &= mask - 1u;
                const int a = r == 0 ? acc[0] : r == 1 ? acc[1] : r == 2 ? acc[2] : acc[3];
                const int t = r == 0 ? thr[0] : r == 1 ? thr[1] : r == 2 ? thr[2] : thr[3];
                emit(qoff + 4 * gq + r, (uint32_t) a, (uint32_t) t, n);
            }
        }
    };
    v4i_t keepA = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff}, keepB = keepA;
    auto take_min = [&](v4i_t &keep, const v4i_t &acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[r] = acc[r] < keep[r] ? acc[r] : keep[r];
    };
    auto publish = [&](const v4i_t &keep, int qoff) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int v = keep[r];
            // the row's minimum on the DPP path (row_shr 1, 2, 4, 8: lane 15 of the row ends up with all sixteen).  (__shfl_xor: its
            // ds_bpermute lane addresses are pure functions of the lane id, were shared with the fused tail's shuffles behind the scan
            // loop and so kept -- spilled -- across it: the 28 bytes per lane of scratch of the <0, true> instantiation.)
#define RII_ROW_MIN_STEP(CTRL) { const int o = __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, 0xf, 0xf, false); v = o < v ? o : v; }
            RII_ROW_MIN_STEP(0x111) RII_ROW_MIN_STEP(0x112) RII_ROW_MIN_STEP(0x114) RII_ROW_MIN_STEP(0x118)
#undef RII_ROW_MIN_STEP
            const int q = qoff + 4 * gq + r, b = qbase + q;
            if (col == 15 && v != 0x7fffffff && b < p.B) atomicMin(&s_thr[q], fs_thr_of((uint32_t) v, s_slk[q]));
        }
    };
    auto adopt = [&](bool first) {
        if (MODE == 0 && tid < 64) {
            // the lane's query, re-derived HERE (round 6): as a loop invariant the two addresses below (8 + 4 bytes per lane) were kept
            // across the scan loop, which has no register to spare -- 24 bytes per lane of scratch (profiles/r05_deep_kernel_stats.txt)
            int q;                                          // (the lane id INSIDE the asm: the builtin form is pure, hoisted and spilled again)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(q));
            const int b = qbase + q;
            if (q < NQ && b < p.B) {
                const uint32_t g = __hip_atomic_load(&p.gthr[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (first) {
                    const uint32_t mine = s_thr[q];
                    if (mine < g) atomicMin(&p.gthr[b], mine);
                }
                atomicMin(&s_thr[q], g);
            }
        }
    };
    auto slow_group = [&](const W *fcp, int64_t gi, bool minima) {
        const W w = fcp[(size_t) gi * 64];
        v4i_t r[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const uint32_t addr = __builtin_amdgcn_perm(C[t], w, fs_mx_sel(t & 3));
            r[t] = *(fs_lds_row_t) (uintptr_t) addr;
        }
        v4i_t accA = {p.bias, p.bias, p.bias, p.bias}, accB = accA;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            const v8i_t ba = __builtin_shufflevector(r[t], r[t + 1], 0, 1, 2, 3, 4, 5, 6, 7);
            accA = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, ba, accA, spidx, 0, 0);
            const v8i_t bb = __builtin_shufflevector(r[4 + t], r[5 + t], 0, 1, 2, 3, 4, 5, 6, 7);
            accB = __builtin_amdgcn_smfmac_i32_16x16x128_i8(spa, bb, accB, spidx, 0, 0);
        }
        const int64_t n = c_begin + gi * 16 + col;
        if (n >= c_end) { accA = v4i_t{0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff}; accB = accA; }
        if (minima) { take_min(keepA, accA); take_min(keepB, accB); }
        else {
            const v4i_t thrA = *reinterpret_cast<const v4i_t *>(s_thr + 4 * gq), thrB = *reinterpret_cast<const v4i_t *>(s_thr + 16 + 4 * gq);
            judge(accA, thrA, (uint32_t) n, 0);
            judge(accB, thrB, (uint32_t) n, 16);
        }
    };
    auto tail = [&](const W *fcp, bool minima) {
        for (int gi = wave; gi < tail_groups; gi += kFsThreads / 64) slow_group(fcp, (int64_t) full * 64 + gi, minima);
    };
    if constexpr (MODE == 0) {
        if (full > 0) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) slow_group(fc, wave * 4 + j, true);
        } else {
            tail(fc, true);
        }
        publish(keepA, 0);
        publish(keepB, 16);
        __syncthreads();
        adopt(true);
        __syncthreads();
    }
    const int step = (MODE == 1) ? p.sample_stride : 1;
    const int ntrip = (MODE == 1) ? (full + step - 1) / step : full;
    auto trip_of = [&](int k) { return (MODE == 0) ? (k + 1 < full ? k + 1 : 0) : k * step; };
    if (ntrip > 0) {
        v4i_t zero4 = {p.bias, p.bias, p.bias, p.bias};
        asm volatile("" : "+v"(zero4));
        const W *pw = fc + (size_t) wave * 4 * 64;
        auto trip_ptr = [&](int k) { return pw + (size_t) trip_of(k < ntrip ? k : ntrip - 1) * 64 * 64; };
        v4i_t ra[8], rb[8];
        W q[4];
        constexpr int S = 64 * (int) sizeof(W);
        {
            const W *p0 = trip_ptr(0), *p1 = trip_ptr(1);
            const W g0 = p0[0], g1 = p0[64];
            fs_mx_issue_hot_dual(g0, C, ra);
            fs_mx_issue_hot_dual(g1, C, rb);
            fs_mx_load<2 * S>(q[2], p0);
            fs_mx_load<3 * S>(q[3], p0);
            fs_mx_load<0>(q[0], p1);
            fs_mx_load<S>(q[1], p1);
        }
        const uint32_t thr_addr = (uint32_t) (lut_bytes + 16 * gq);
        v4i_t thrA, thrB;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(thrA), "=&v"(thrB) : "v"(thr_addr));
        for (int k = 0; k < ntrip; ++k) {
            const int it = trip_of(k);
            const W *pn = trip_ptr(k + 1), *pnn = trip_ptr(k + 2);
            const uint32_t n = (uint32_t) c_begin + (uint32_t) it * kFsThreads + wave * 64 + col;
            v4i_t accA, accB;
            fs_mx_vmwait<3>(q[2]);
            fs_mx_wait<8>(ra);                                    // group 0 (the 8 younger reads are group 1's)
            fs_mx_reduce_refill_dual(ra, q[2], C, spa, spidx, zero4, accA, accB);
            fs_mx_load<2 * S>(q[2], pn);
            if (MODE == 1) { take_min(keepA, accA); take_min(keepB, accB); } else { judge(accA, thrA, n, 0); judge(accB, thrB, n, 16); }
            fs_mx_vmwait<3>(q[3]);
            fs_mx_wait<8>(rb);                                    // group 1
            fs_mx_reduce_refill_dual(rb, q[3], C, spa, spidx, zero4, accA, accB);
            fs_mx_load<3 * S>(q[3], pn);
            if (MODE == 1) { take_min(keepA, accA); take_min(keepB, accB); } else { judge(accA, thrA, n + 16, 0); judge(accB, thrB, n + 16, 16); }
            // thresholds: re-read once per trip into the live registers (any mix of old and new words is a valid set)
            if constexpr (MODE == 0) {
                asm volatile("ds_read_b128 %0, %1 ; rii:inflight-ok (tools/check_isa_inflight.py)" : "+v"(thrA) : "v"(thr_addr));
                asm volatile("ds_read_b128 %0, %1 offset:64 ; rii:inflight-ok (tools/check_isa_inflight.py)" : "+v"(thrB) : "v"(thr_addr));
            }
            fs_mx_vmwait<3>(q[0]);
            fs_mx_wait<(MODE == 0) ? 10 : 8>(ra);                 // group 2 (younger: group 3's rows and the two threshold reads)
            fs_mx_reduce_refill_dual(ra, q[0], C, spa, spidx, zero4, accA, accB);
            fs_mx_load<0>(q[0], pnn);
            if (MODE == 1) { take_min(keepA, accA); take_min(keepB, accB); } else { judge(accA, thrA, n + 32, 0); judge(accB, thrB, n + 32, 16); }
            fs_mx_vmwait<3>(q[1]);
            fs_mx_wait<8>(rb);                                    // group 3 (and the thresholds: older than group 2's refills)
            fs_mx_reduce_refill_dual(rb, q[1], C, spa, spidx, zero4, accA, accB);
            fs_mx_load<S>(q[1], pnn);
            if (MODE == 1) { take_min(keepA, accA); take_min(keepB, accB); } else { judge(accA, thrA, n + 48, 0); judge(accB, thrB, n + 48, 16); }
            adopt(false);
        }
        fs_mx_wait<0>(ra);
        fs_mx_wait<0>(rb);
        fs_mx_vmwait<0>(q[0]); fs_mx_vmwait<0>(q[1]); fs_mx_vmwait<0>(q[2]); fs_mx_vmwait<0>(q[3]);
    }
    {
        // the lane's code pointer re-derived behind the scan loop (round 6): kept live across it, it was spilled (see adopt)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        tail(reinterpret_cast<const W *>(p.codes) + (size_t) (c_begin / 16) * 64 + ln, MODE == 1);
    }

    if (MODE == 0 && p.lcap > 0) {
        __syncthreads();
        if (tid < NQ) {
            const int b = qbase + tid;
            const unsigned int c = min(s_lcnt[tid], (unsigned int) p.lcap);
            s_lcnt[NQ + tid] = (c && b < p.B) ? atomicAdd(&p.cand_count[b], c) : 0u;
        }
        __syncthreads();
        for (int q = tid >> 6; q < NQ; q += kFsThreads >> 6) {        // one wave per query
            const int b = qbase + q;
            const unsigned int c = min(s_lcnt[q], (unsigned int) p.lcap), base = s_lcnt[NQ + q];
            for (unsigned int i = tid & 63; i < c; i += 64)
                if (base + i < (unsigned int) p.cap) {
                    if constexpr (TAIL) fs_cand_store(&p.cand[(size_t) b * p.cap + base + i], s_lcand[(size_t) q * p.lcap + i]);     // write-through: read by the tile's last block
                    else p.cand[(size_t) b * p.cap + base + i] = s_lcand[(size_t) q * p.lcap + i];
                }
        }
    }
    if constexpr (TAIL) {
        fs_kernarg_t pa = (fs_kernarg_t) __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pa));
        bool spilled = false;
        for (int q = 0; q < NQ; ++q) spilled |= s_lcnt[q] > (uint32_t) pa->lcap;
        int t2 = tid;                                          // (opaque: what the tail derives from the thread index is derived behind the
        asm volatile("" : "+v"(t2));                           //  scan loop, not kept -- and spilled -- across it)
        if (fs_tail_arrive(pa, s_lcnt, t2, spilled)) {         // the last chunk-block of the tile pair re-ranks its 32 queries
            if (pa->tail.Ds == 4) fs_tail_rerank<float4, 32>(pa, qbase, smem, t2);
            else fs_tail_rerank<float2, 32>(pa, qbase, smem, t2);
        }
    }
    if constexpr (MODE == 1) {
        const size_t G = (size_t) gridDim.x * kFsMxSeg;
        const size_t seg = (size_t) blockIdx.x * kFsMxSeg + wave * 16 + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bA = qbase + 4 * gq + r, bB = bA + 16;
            if (bA < p.B) p.segmin[(size_t) bA * G + seg] = (uint16_t) (keepA[r] > 0xffff ? 0xffff : keepA[r]);
            if (bB < p.B) p.segmin[(size_t) bB * G + seg] = (uint16_t) (keepB[r] > 0xffff ? 0xffff : keepB[r]);
        }
    }
}

template <int MODE, bool TAIL = false> static hipError_t launch_fscan_mx_dual_t(const FsArgs &a, int chunks, hipStream_t st)
{
    const size_t tab = (size_t) 2 * 16 * 256 * 16 + 32 * 4 + 32 * 8 + 32 * 4;
    FsArgs b = a;
    b.lcap = (MODE != 0) ? 0 : (int) std::min<size_t>(128, (kFsLdsBytes - tab) / ((size_t) 32 * 8));
    size_t smem = tab + (size_t) 32 * 8 * b.lcap;
    if (TAIL) smem = std::max(smem, kFsTailLds);
    auto kern = fscan_mx_dual_kernel<MODE, TAIL>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    launch_timed(kern, dim3(chunks, (a.B + 31) / 32), dim3(kFsThreads), smem, st, b);
    return hipGetLastError();
}

template <int T, int MODE, int QR = 16, bool TAIL = false, bool PIPE = false> static hipError_t launch_fscan_mx_t(const FsArgs &a, int chunks, int tiles, hipStream_t st)
{
    const size_t tab = (size_t) a.M * a.Ks * QR + 64 + (size_t) QR * 8 + 64;
    FsArgs b = a;
    b.lcap = (MODE != 0) ? 0 : (int) std::min<size_t>(128, (kFsLdsBytes - tab) / ((size_t) QR * 8));
    size_t smem = tab + (size_t) QR * 8 * b.lcap;
    if (TAIL) smem = std::max(smem, kFsTailLds);
    auto kern = fscan_mx_kernel<T, MODE, QR, TAIL, PIPE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    launch_timed(kern, dim3(chunks, tiles), dim3(kFsThreads), smem, st, b);
    return hipGetLastError();
}

template <int MODE> static hipError_t launch_fscan_mode(const FsArgs &a, int chunks, bool rot, bool mx, hipStream_t st)
{
    const int qr = fastscan_rows(a.M, a.Ks);
    const int tiles = (a.B + qr - 1) / qr;
    if (rot && mx) {
        if constexpr (MODE == 0) {
            if (a.tail.queries) {             // top-1 with the re-rank folded into the scan's tail
                if (a.M == 16 && a.dual) return launch_fscan_mx_dual_t<0, true>(a, chunks, st);
                if (a.M == 16) return launch_fscan_mx_t<4, 0, 16, true>(a, chunks, tiles, st);
                if (a.M == 32) return launch_fscan_mx_t<8, 0, 16, true>(a, chunks, tiles, st);
                return hipErrorInvalidValue;
            }
            // unsigned table bytes (63 / 127 levels): the sums of a group are judged one group late, behind the next group's matrix
            // instructions (PIPE); signed bytes (255 levels: accumulators start at 128 M) keep the in-order judge
            if (a.bias == 0 && a.pipe) {
                if (a.M == 16 && !a.dual) return launch_fscan_mx_t<4, 0, 16, false, true>(a, chunks, tiles, st);
                if (a.M == 32) return launch_fscan_mx_t<8, 0, 16, false, true>(a, chunks, tiles, st);
            }
        }
        if (a.M == 16 && a.dual) return launch_fscan_mx_dual_t<MODE>(a, chunks, st);
        if (a.M == 16) return launch_fscan_mx_t<4, MODE>(a, chunks, tiles, st);
        if (a.M == 32) return launch_fscan_mx_t<8, MODE>(a, chunks, tiles, st);
        if (a.M == 64) return launch_fscan_mx_t<16, MODE, 8>(a, chunks, tiles, st);
        return hipErrorInvalidValue;
    }
    if (rot) {
        if (qr == 16 && a.M == 16) return launch_fscan_t<4, 256, MODE, 16, true>(a, chunks, tiles, st);
        if (qr == 16 && a.M == 32) return launch_fscan_t<8, 256, MODE, 16, true>(a, chunks, tiles, st);
        return hipErrorInvalidValue;
    }
    if (qr == 16) {
        if (a.Ks == 256 && a.M == 8) return launch_fscan_t<2, 256, MODE, 16>(a, chunks, tiles, st);
        if (a.Ks == 256 && a.M == 16) return launch_fscan_t<4, 256, MODE, 16>(a, chunks, tiles, st);
        if (a.Ks == 256 && a.M == 32) return launch_fscan_t<8, 256, MODE, 16>(a, chunks, tiles, st);
        return launch_fscan_t<0, 0, MODE, 16>(a, chunks, tiles, st);
    }
    if (a.Ks == 256 && a.M == 64) return launch_fscan_t<16, 256, MODE, 8>(a, chunks, tiles, st);
    return launch_fscan_t<0, 0, MODE, 8>(a, chunks, tiles, st);
}

hipError_t launch_fscan(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const uint8_t *d_qlut,
                        const int32_t *d_slack, int B, int chunks, int64_t chunk_len, unsigned long long *d_cand,
                        unsigned int *d_cand_count, int cap, int mode, uint16_t *d_segmin, const uint32_t *d_thr16,
                        uint32_t *d_gthr, int sample_stride, int mx, hipStream_t st, int quarter, int dual, int levels, const FsTail *tail, int pipe)
{
    // d_codes: formatted lookups (launch_fcodes_format, same `mx`) for fs_rot_supported shapes, the plain codes otherwise
    if (B == 0 || n_codes == 0) return hipSuccess;
    const bool rot = fs_rot_supported(M, Ks, mx);
    FsArgs a;
    a.quarter = quarter;
    a.dual = ((dual & 1) && mx && M == 16 && rot) ? 1 : 0;
    a.bias = levels > 127 ? 128 * M : 0;
    a.pipe = pipe;
    if (tail && tail->queries) {
        if (mode != 0 || !fscan_tail_supported(M, Ks, tail->Ds, mx)) return hipErrorInvalidValue;
        a.tail = *tail;
    }
    a.gthr = d_gthr;
    a.sample_stride = sample_stride < 1 ? 1 : sample_stride;
    a.codes = d_codes; a.n_codes = n_codes; a.M = M; a.Ks = Ks; a.qlut = d_qlut; a.slack = d_slack; a.B = B;
    a.chunk_len = chunk_len; a.cand = d_cand; a.cand_count = d_cand_count; a.cap = cap; a.segmin = d_segmin;
    a.thr16 = d_thr16;
    if (mode == 1) return launch_fscan_mode<1>(a, chunks, rot, mx != 0, st);
    if (mode == 2) return launch_fscan_mode<2>(a, chunks, rot, mx != 0, st);
    return launch_fscan_mode<0>(a, chunks, rot, mx != 0, st);
}

bool fscan_tail_supported(int M, int Ks, int Ds, int mx)
{
    return mx && Ks == 256 && (M == 16 || M == 32) && (Ds == 4 || Ds == 2) && fs_rot_supported(M, Ks, mx);
}
int fscan_tail_flags(int M, int Ks, int mx, int dual, int64_t B)
{
    const int qpb = fscan_queries_per_block(M, Ks, mx, dual);
    return (int) ((B + qpb - 1) / qpb);
}
int fscan_queries_per_block(int M, int Ks, int mx, int dual)
{
    return (dual && mx && M == 16 && fs_rot_supported(M, Ks, mx)) ? 32 : fastscan_rows(M, Ks);
}
int fscan_mx_subspace(int M, int lane, int t)
{
    return M == 64 ? fs_mx_subspace64(lane >> 4, lane & 15, t) : fs_mx_subspace(lane >> 4, lane & 15, t);
}
// lane segments per chunk and query of the MODE 1 pass (the G of launch_kth_threshold is chunks times this)
int fscan_segments_per_chunk(int M, int Ks, int mx) { return (mx && fs_rot_supported(M, Ks, mx)) ? kFsMxSeg : kFsThreads; }
// bytes of the formatted copy of n codes: fscan_mx_kernel's is a permutation of the code bytes written in whole groups of 16
// codes, fscan_kernel's holds a 16-bit (half, ks, slot) value per code byte
int64_t fcodes_bytes(int64_t n, int M, int mx) { return mx ? (n + 15) / 16 * 16 * M : n * M * 2;
}

// ---------------------------------------------------------------------------------------------------
// top-k, between the passes: v_k = k-th smallest of the G segment minima of query b  ->  thr16[b] = v_k + slack + 1.
// One block per query; counting select over the value range [0, M*63].
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kth_threshold_kernel(const uint16_t *__restrict__ segmin, int64_t G, int k,
                                                            int maxv, const int32_t *__restrict__ slack,
                                                            uint32_t *__restrict__ thr16)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *hist = reinterpret_cast<int *>(smem);                 // [maxv + 1]
    __shared__ int s_part[256];
    __shared__ int s_res;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const int nb = maxv + 1;
    for (int i = tid; i < nb; i += 256) hist[i] = 0;
    if (tid == 0) s_res = -1;
    __syncthreads();
    const uint16_t *row = segmin + (size_t) b * G;
    for (int64_t i = tid; i < G; i += 256) {
        const int v = row[i];
        if (v <= maxv) atomicAdd(&hist[v], 1);
    }
    __syncthreads();
    const int per = (nb + 255) / 256;
    const int lo = tid * per, hi = (lo + per < nb) ? lo + per : nb;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += hist[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int acc = 0, owner = -1;
        for (int t = 0; t < 256; ++t) {
            if (acc + s_part[t] >= k) { owner = t; break; }
            acc += s_part[t];
        }
        s_part[0] = acc;                 // count before the owner's range (reuse slot 0 after reading all)
        s_res = owner;
    }
    __syncthreads();
    const int owner = s_res;
    if (owner < 0) {
        if (tid == 0) thr16[b] = 0xffffu;                       // fewer than k non-empty segments: keep everything
        return;
    }
    if (tid == owner) {
        int acc = s_part[0];
        int v = hi - 1;
        for (int i = lo; i < hi; ++i) {
            acc += hist[i];
            if (acc >= k) { v = i; break; }
        }
        const uint32_t t = (uint32_t) v + (uint32_t) slack[b] + 1u;
        thr16[b] = t > 0xffffu ? 0xffffu : t;
    }
}

hipError_t launch_kth_threshold(const uint16_t *d_segmin, int64_t G, int64_t B, int k, int maxv, const int32_t *d_slack,
                                uint32_t *d_thr16, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    const size_t smem = (size_t) (maxv + 1) * sizeof(int);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kth_threshold_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kth_threshold_kernel, dim3((unsigned) B), dim3(256), smem, st, d_segmin, G, k, maxv, d_slack,
                       d_thr16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// stage 2: exact re-rank.  One block (256 threads) per query, exact fp32 table of that query in LDS.
// If the candidate buffer overflowed (pathological: e.g. thousands of duplicate nearest codes) the block falls
// back to an exact scan of all codes for its query -- still on the GPU, still exact.
// ---------------------------------------------------------------------------------------------------
struct RrArgs {
    const uint8_t *codes;
    int64_t n_codes;
    int M, Ks;
    const float *lut;
    int QT;
    const int32_t *slack;
    const unsigned long long *cand;
    const unsigned int *cand_count;
    int cap;
    const int64_t *remap;
    const int32_t *perm;          // scan position -> code id (scanorder.hip), or NULL when the codes are in id order
    int64_t *out_ids;
    float *out_dists;
    int topk;
    int indirect = 0;                // 1: `codes` is the whole database and position n stands for the code remap[n] (subset search)
    int32_t *flag_list = nullptr;    // top-k: queries whose k+1 smallest distances hold an exact tie (redone by tieorder.hip)
    int *nflag = nullptr;
    unsigned int *peak = nullptr;    // top-1 (round 6): running maximum of the candidate counts -- the engine sizes the next batch's buffers by it
};

__global__ __launch_bounds__(256) void rerank_top1_kernel(RrArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const int MK = p.M * p.Ks;
    unsigned long long *red = reinterpret_cast<unsigned long long *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    {
        const float *src = p.lut + (size_t) (b / p.QT) * MK * p.QT + (b % p.QT);
        if (p.QT == 1 && (MK & 3) == 0) {               // plain [b][M*Ks] layout (what the fused table kernel writes): 16-byte copies
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(lds);
#pragma unroll 4
            for (int i = tid; i < MK / 4; i += blockDim.x) d4[i] = s4[i];
        } else {
            for (int i = tid; i < MK; i += blockDim.x) lds[i] = src[(size_t) i * p.QT];
        }
        if (tid == 0) { red[0] = ~0ull; red[1] = ~0ull; }
    }
    __syncthreads();
    const unsigned int cnt = p.cand_count[b];
    if (tid == 0 && p.peak && cnt > (unsigned int) p.cap / 2) atomicMax(p.peak, cnt);     // (rare: only lists that come near the buffer's end)
    unsigned long long best = ~0ull;
    if (cnt <= (unsigned int) p.cap) {
        const unsigned long long *cand = p.cand + (size_t) b * p.cap;
        // global minimum of the quantised sums (the arg-min code is always among the candidates)
        unsigned long long amin = ~0ull;
        for (unsigned int i = tid; i < cnt; i += blockDim.x) {
            const unsigned long long a = cand[i] >> 32;
            amin = a < amin ? a : amin;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(amin, off);
            amin = o < amin ? o : amin;
        }
        if ((tid & 63) == 0) atomicMin(&red[0], amin);
        __syncthreads();
        const unsigned long long lim = red[0] + (unsigned long long) (uint32_t) p.slack[b];
        if (p.M == 16 && p.Ks == 256 && !p.indirect) {
            // round 6: long lists (a structured Deep-shaped set leaves ~13 k candidates per query, profiles/r06_deep_structured_levels.json):
            // four candidates per thread with their 16-byte code rows in flight together -- one at a time, each row was a dependent
            // round trip behind its record (1.4 ms of that 7 ms step)
            for (unsigned int i0 = tid; i0 < cnt; i0 += 4 * blockDim.x) {
                unsigned long long c[4];
                uint4 row[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned int i = i0 + u * blockDim.x;
                    c[u] = i < cnt ? cand[i] : ~0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool live = c[u] != ~0ull && (c[u] >> 32) <= lim;
                    if (!live) c[u] = ~0ull;
                    row[u] = *reinterpret_cast<const uint4 *>(p.codes + (size_t) (live ? (uint32_t) (c[u] & 0xffffffffu) : 0u) * 16);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c[u] == ~0ull) continue;
                    const uint32_t wds[4] = {row[u].x, row[u].y, row[u].z, row[u].w};
                    float d = 0.f;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int j = 0; j < 4; ++j) d = __fadd_rn(d, lds[(a * 4 + j) * 256 + ((wds[a] >> (8 * j)) & 0xffu)]);
                    const uint32_t n = (uint32_t) (c[u] & 0xffffffffu);
                    const uint32_t id = p.perm ? (uint32_t) p.perm[n] : n;
                    const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(d)) << 32) | id;
                    best = key < best ? key : best;
                }
            }
        } else
        for (unsigned int i = tid; i < cnt; i += blockDim.x) {
            const unsigned long long c = cand[i];
            if ((c >> 32) > lim) continue;
            const uint32_t n = (uint32_t) (c & 0xffffffffu);
            const float d = exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[n] : (int64_t) n) * p.M, p.M, p.Ks);
            const uint32_t id = p.perm ? (uint32_t) p.perm[n] : n;
            const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(d)) << 32) | id;
            best = key < best ? key : best;
        }
    } else {
        for (int64_t n = tid; n < p.n_codes; n += blockDim.x) {
            const float d = exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[n] : (int64_t) n) * p.M, p.M, p.Ks);
            const uint32_t id = p.perm ? (uint32_t) p.perm[n] : (uint32_t) n;
            const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(d)) << 32) | id;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if ((tid & 63) == 0 && best != ~0ull) atomicMin(&red[1], best);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long k = red[1];
        const uint32_t idx = (uint32_t) (k & 0xffffffffu);
        p.out_ids[b * p.topk] = (k == ~0ull) ? -1 : (p.remap ? p.remap[idx] : (int64_t) idx);
        p.out_dists[b * p.topk] = (k == ~0ull) ? INFINITY : __uint_as_float(f32_unorderable((uint32_t) (k >> 32)));
    }
}

// The same without a table in global memory (round 3): the candidates of a query (~150 of 1 M codes at the bench shape) touch at
// most cnt * M of the M * Ks table entries, and each entry is one fvec_L2sqr of the query's sub-vector with a codeword of the L2-
// resident codebook -- the very expression the table kernels evaluate, so the distances are bit-identical.  One thread per
// candidate, eight codeword loads in flight per thread, additions in the reference's m order.  Saves the 32 MB fp32 table the
// table kernel would write and this kernel would read back per 1024-query batch.  Ks = 256, M a multiple of 8, Ds = 4 / 2.
template <typename Vec>
__global__ __launch_bounds__(256) void rerank_top1_direct_kernel(RrArgs p, const float *__restrict__ queries,
                                                                 const float *__restrict__ codewords)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int Ds = (int) (sizeof(Vec) / sizeof(float));
    float *lds = reinterpret_cast<float *>(smem);                    // [M * 256] exact table: built only if the candidates overflowed
    const int MK = p.M * 256;
    unsigned long long *red = reinterpret_cast<unsigned long long *>(smem + (size_t) MK * 4);
    Vec *s_q = reinterpret_cast<Vec *>(red + 2);                     // [M] the query's sub-vectors
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const Vec *cw = reinterpret_cast<const Vec *>(codewords);
    if (tid < p.M) s_q[tid] = reinterpret_cast<const Vec *>(queries + b * (int64_t) (p.M * Ds))[tid];
    if (tid == 0) { red[0] = ~0ull; red[1] = ~0ull; }
    __syncthreads();
    const unsigned int cnt = p.cand_count[b];
    unsigned long long best = ~0ull;
    if (cnt <= (unsigned int) p.cap) {
        const unsigned long long *cand = p.cand + (size_t) b * p.cap;
        // candidates emitted under early (loose) thresholds can be pruned with the final minimum of the quantised sums -- worth a
        // pass over the list and a barrier only when there is more than one candidate per thread (40 per query at the bench shape)
        unsigned long long lim = ~0ull;
        if (cnt > blockDim.x) {
            unsigned long long amin = ~0ull;
            for (unsigned int i = tid; i < cnt; i += blockDim.x) {
                const unsigned long long a = cand[i] >> 32;
                amin = a < amin ? a : amin;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(amin, off);
                amin = o < amin ? o : amin;
            }
            if ((tid & 63) == 0) atomicMin(&red[0], amin);
            __syncthreads();
            lim = red[0] + (unsigned long long) (uint32_t) p.slack[b];
        }
        for (unsigned int i = tid; i < cnt; i += blockDim.x) {
            const unsigned long long c = cand[i];
            if ((c >> 32) > lim) continue;
            const uint32_t n = (uint32_t) (c & 0xffffffffu);
            const uint2 *code = reinterpret_cast<const uint2 *>(p.codes + (size_t) (p.indirect ? (int64_t) p.remap[n] : (int64_t) n) * p.M);
            float dist = 0.f;
            for (int m0 = 0; m0 < p.M; m0 += 16) {                   // 16 codeword loads in flight per thread: one or two round trips per code
                const uint2 wa = code[m0 >> 3];
                const uint2 wb = (m0 + 8 < p.M) ? code[(m0 >> 3) + 1] : make_uint2(0u, 0u);
                const uint32_t wd[4] = {wa.x, wa.y, wb.x, wb.y};
                Vec cv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t ks = (wd[j >> 2] >> (8 * (j & 3))) & 0xffu;
                    cv[j] = cw[(m0 + j < p.M ? m0 + j : 0) * 256 + ks];
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (m0 + j < p.M) dist = __fadd_rn(dist, fvec_l2sqr_vec(s_q[m0 + j], cv[j]));
            }
            const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(dist)) << 32) | n;
            best = key < best ? key : best;
        }
    } else {
        // the candidate buffer overflowed (e.g. thousands of duplicated nearest codes): exact table in LDS, every code scanned
        for (int i = tid; i < MK; i += blockDim.x) lds[i] = fvec_l2sqr_vec(s_q[i >> 8], cw[i]);
        __syncthreads();
        for (int64_t n = tid; n < p.n_codes; n += blockDim.x) {
            const float d = exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[n] : (int64_t) n) * p.M, p.M, 256);
            const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(d)) << 32) | (uint32_t) n;
            best = key < best ? key : best;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o < best ? o : best;
    }
    if ((tid & 63) == 0 && best != ~0ull) atomicMin(&red[1], best);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long k = red[1];
        const uint32_t idx = (uint32_t) (k & 0xffffffffu);
        p.out_ids[b * p.topk] = (k == ~0ull) ? -1 : (p.remap ? p.remap[idx] : (int64_t) idx);
        p.out_dists[b * p.topk] = (k == ~0ull) ? INFINITY : __uint_as_float(f32_unorderable((uint32_t) (k >> 32)));
    }
}

bool rerank_direct_supported(int M, int Ks, int Ds) { return Ks == 256 && (M % 8) == 0 && M <= 64 && (Ds == 4 || Ds == 2); }
hipError_t launch_rerank_top1_direct(const uint8_t *d_codes, int64_t n_codes, int M, int Ds, const float *d_queries,
                                     const float *d_codewords, const int32_t *d_slack, const unsigned long long *d_cand,
                                     const unsigned int *d_cand_count, int cap, const int64_t *d_remap, int64_t B,
                                     int64_t *d_out_ids, float *d_out_dists, int topk, int indirect, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    RrArgs a;
    a.indirect = indirect;
    a.perm = nullptr;
    a.codes = d_codes; a.n_codes = n_codes; a.M = M; a.Ks = 256; a.lut = nullptr; a.QT = 1; a.slack = d_slack;
    a.cand = d_cand; a.cand_count = d_cand_count; a.cap = cap; a.remap = d_remap; a.out_ids = d_out_ids;
    a.out_dists = d_out_dists; a.topk = topk;
    const size_t smem = (size_t) M * 256 * sizeof(float) + 16 + (size_t) M * 16;
    if (Ds == 4) {
        auto kern = rerank_top1_direct_kernel<float4>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned) B), dim3(256), smem, st, a, d_queries, d_codewords);
    } else {
        auto kern = rerank_top1_direct_kernel<float2>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned) B), dim3(256), smem, st, a, d_queries, d_codewords);
    }
    return hipGetLastError();
}

hipError_t launch_rerank_top1(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const float *d_lut, int QT,
                              const int32_t *d_slack, const unsigned long long *d_cand,
                              const unsigned int *d_cand_count, int cap, const int64_t *d_remap,
                              const int32_t *d_perm, int64_t B, int64_t *d_out_ids, float *d_out_dists, int topk,
                              int indirect, hipStream_t st, unsigned int *d_peak)
{
    if (B == 0) return hipSuccess;
    RrArgs a;
    a.indirect = indirect;
    a.perm = d_perm;
    a.peak = d_peak;
    a.codes = d_codes; a.n_codes = n_codes; a.M = M; a.Ks = Ks; a.lut = d_lut; a.QT = QT; a.slack = d_slack;
    a.cand = d_cand; a.cand_count = d_cand_count; a.cap = cap; a.remap = d_remap; a.out_ids = d_out_ids;
    a.out_dists = d_out_dists; a.topk = topk;
    const size_t smem = (((size_t) M * Ks * sizeof(float) + 15) & ~(size_t) 15) + 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rerank_top1_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rerank_top1_kernel, dim3((unsigned) B), dim3(256), smem, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// stage 2 for top-k (k > 1): exact distances of every candidate (or of every code if the candidate buffer overflowed)
// streamed through a block-local top-k: keys (orderable dist << 32 | index) below the current k-th best are appended
// to an LDS buffer, which is bitonic-sorted and cut back to k whenever it could overflow.  Output in (dist, id) order.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rerank_topk_kernel(RrArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const int MK = p.M * p.Ks;
    unsigned long long *buf = reinterpret_cast<unsigned long long *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));
    unsigned long long &s_thr = buf[kRrBuf];                                   // all LDS in the dynamic region
    unsigned int &s_cnt = *reinterpret_cast<unsigned int *>(&buf[kRrBuf + 1]);
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const int k = p.topk;
    {
        const float *src = p.lut + (size_t) (b / p.QT) * MK * p.QT + (b % p.QT);
        if (p.QT == 1 && (MK & 3) == 0) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(lds);
#pragma unroll 4
            for (int i = tid; i < MK / 4; i += 256) d4[i] = s4[i];
        } else {
            for (int i = tid; i < MK; i += 256) lds[i] = src[(size_t) i * p.QT];
        }
        if (tid == 0) { s_cnt = 0u; s_thr = ~0ull; }
    }
    const unsigned int ncand = p.cand_count[b];
    const bool overflow = ncand > (unsigned int) p.cap;
    const int64_t total = overflow ? p.n_codes : (int64_t) ncand;
    // the k+1 smallest keys are tracked: two equal distances among them mean the reference's answer hinges on
    // std::partial_sort's heap order (tieorder.hip redoes the query).  Every code tied with the k-th distance is a
    // candidate (see the exactness proof above), so a tie across the cut cannot hide outside the candidate set.
    const int k1 = (int64_t) k + 1 < total ? k + 1 : (int) total;
    const unsigned long long *cand = p.cand + (size_t) b * p.cap;
    for (int64_t base = 0; base < total; base += 256) {
        // the make-room decision must be uniform (the branch holds barriers): snapshot the counter between two barriers,
        // after every append of the previous trip and before any append of this one
        __syncthreads();
        const unsigned int cnt_now = s_cnt;
        __syncthreads();
        if (cnt_now + 256u > (unsigned int) kRrBuf) {
            for (int i = tid; i < kRrBuf; i += 256)
                if ((unsigned int) i >= cnt_now) buf[i] = ~0ull;
            rr_bitonic_sort(buf, tid);
            if (tid == 0) { s_cnt = (unsigned int) k1; s_thr = buf[k1 - 1]; }
            __syncthreads();
        }
        const int64_t i = base + tid;
        if (i < total) {
            const uint32_t n = overflow ? (uint32_t) i : (uint32_t) (cand[i] & 0xffffffffu);
            const float d = exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[n] : (int64_t) n) * p.M, p.M, p.Ks);
            const uint32_t id = p.perm ? (uint32_t) p.perm[n] : n;
            const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(d)) << 32) | id;
            if (key < s_thr) buf[atomicAdd(&s_cnt, 1u)] = key;
        }
    }
    __syncthreads();
    int nsort = 64;                               // smallest power of two covering the keys actually collected
    while (nsort < (int) s_cnt || nsort < k) nsort <<= 1;
    for (int i = tid; i < nsort; i += 256)
        if ((unsigned int) i >= s_cnt) buf[i] = ~0ull;
    rr_bitonic_sort(buf, tid, nsort);
    int tie = 0;
    for (int j = tid; j + 1 < k1; j += 256)
        if ((buf[j] >> 32) == (buf[j + 1] >> 32)) tie = 1;
    if (__syncthreads_or(tie) && p.flag_list) {
        if (tid == 0) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) b;
        return;                                    // linear_tie_kernel writes this row
    }
    for (int j = tid; j < k; j += 256) {
        const unsigned long long key = buf[j];
        const uint32_t idx = (uint32_t) (key & 0xffffffffu);
        p.out_ids[b * k + j] = (key == ~0ull) ? -1 : (p.remap ? p.remap[idx] : (int64_t) idx);
        p.out_dists[b * k + j] = (key == ~0ull) ? INFINITY : __uint_as_float(f32_unorderable((uint32_t) (key >> 32)));
    }
}

int rerank_topk_max_k() { return kRrBuf / 2 - 1; }      // the k+1 smallest keys are tracked

hipError_t launch_rerank_topk(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const float *d_lut, int QT,
                              const unsigned long long *d_cand, const unsigned int *d_cand_count, int cap,
                              const int64_t *d_remap, const int32_t *d_perm, int64_t B, int64_t *d_out_ids,
                              float *d_out_dists, int topk, int32_t *d_flag_list, int *d_nflag, int indirect, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    RrArgs a;
    a.indirect = indirect;
    a.flag_list = d_flag_list; a.nflag = d_nflag;
    a.perm = d_perm;
    a.codes = d_codes; a.n_codes = n_codes; a.M = M; a.Ks = Ks; a.lut = d_lut; a.QT = QT; a.slack = nullptr;
    a.cand = d_cand; a.cand_count = d_cand_count; a.cap = cap; a.remap = d_remap; a.out_ids = d_out_ids;
    a.out_dists = d_out_dists; a.topk = topk;
    const size_t smem = (((size_t) M * Ks * sizeof(float) + 15) & ~(size_t) 15) + (size_t) (kRrBuf + 2) * 8;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rerank_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rerank_topk_kernel, dim3((unsigned) B), dim3(256), smem, st, a);
    return hipGetLastError();
}

}  // namespace riiamd
