// kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the IVFPQ query hot path.
//
// Reference functions restated on the GPU (paths relative to the reference repo):
//   lut_build_kernel        fvec_L2sqr (src/distance.h:117-252) inside RiiCpp::DTable (src/rii.h:361-373)
//   scan_kernel             RiiCpp::ADist over all codes + top-1 (src/rii.h:195-242, 386-394)
//   ivf_*_kernel            RiiCpp::QueryIvf (src/rii.h:244-326) incl. libstdc++'s std::partial_sort order
//   symtab/assign kernels   PQKMeans tables + predict_one (src/pqkmeans.cpp:23-42,152-173,193-218) as used by
//                           RiiCpp::UpdatePostingLists (src/rii.h:335-359)
//   pqk_hist/vote kernels   PQKMeans::fit centre update (src/pqkmeans.cpp:109-123,223-260)
//
// Numerics contract: every fp32 operation that decides a result is written with an explicit
// round-to-nearest intrinsic (__fadd_rn/__fsub_rn/__fmul_rn/__fmaf_rn) in the reference's order, and the
// file is compiled with -ffp-contract=off, so distances are bit-identical to the reference's.
//
// Why the scan is not a GEMM: see DESIGN.md.  The scan is an LDS-gather kernel: the M x Ks table of QT
// queries lives in LDS as [m][ks][QT] so that ONE ds_read_b128 returns the entries of four queries for one
// code byte; codes stream from HBM/L2 as coalesced 16-byte loads; accumulation is sequential over m.
#include <type_traits>
#include "rii_internal.h"
#include "rii_device.h"
#include <float.h>
#include <algorithm>

namespace riiamd {

int lut_tile_for(int M, int Ks)
{
    const size_t one = (size_t) M * Ks * sizeof(float);
    if (one * 4 <= (size_t) kMaxLutLdsBytes) return 4;
    if (one * 2 <= (size_t) kMaxLutLdsBytes) return 2;
    if (one <= (size_t) kMaxLutLdsBytes) return 1;
    return 0;
}

// ===================================================================================================
// (a1,a2) distance-table build, exact VALU path.  One thread per (query, m, ks).
// ===================================================================================================
__global__ __launch_bounds__(256) void lut_build_kernel(const float *__restrict__ queries, int64_t B,
                                                        const float *__restrict__ codewords, int M, int Ks,
                                                        int Ds, int arch, int QT, float *__restrict__ lut)
{
    const int MK = M * Ks;
    const int64_t total = B * (int64_t) MK;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t b = t / MK;
        const int i = (int) (t - b * MK);
        const int m = i / Ks;
        const float *q = queries + b * (int64_t) (M * Ds) + (int64_t) m * Ds;
        const float *c = codewords + (size_t) i * Ds;
        lut[lut_index(b, i, MK, QT)] = fvec_l2sqr_any(q, c, Ds, arch);
    }
}

hipError_t launch_lut_build(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ks, int Ds,
                            int arch, int QT, float *d_lut, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    const int64_t total = B * (int64_t) M * Ks;
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(lut_build_kernel, dim3(blocks), dim3(256), 0, st, d_queries, B, d_codewords, M, Ks, Ds,
                       arch, QT, d_lut);
    return hipGetLastError();
}

// tile-interleaved -> [B][M][Ks] (only for the rii_dtable() debug/parity entry point)
__global__ void lut_untile_kernel(const float *__restrict__ lut, int64_t B, int MK, int QT, float *__restrict__ out)
{
    const int64_t total = B * (int64_t) MK;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t b = t / MK;
        const int i = (int) (t - b * MK);
        out[t] = lut[lut_index(b, i, MK, QT)];
    }
}
hipError_t launch_lut_untile(const float *d_lut, int64_t B, int M, int Ks, int QT, float *d_out, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    const int64_t total = B * (int64_t) M * Ks;
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(lut_untile_kernel, dim3(blocks), dim3(256), 0, st, d_lut, B, M * Ks, QT, d_out);
    return hipGetLastError();
}

// ===================================================================================================
// (a2, measured variant) distance-table build on the matrix cores: T = |q_m|^2 - 2 q_m.c + |c|^2 with the
// cross term as one v_mfma_f32_16x16x4_f32 per 16 queries x 16 codewords x 4 dims.  Not bit-identical to
// the reference (different rounding points): opt-in through option "lut_mode" = RII_LUT_MFMA.
// ===================================================================================================
__global__ void codeword_norms_kernel(const float *__restrict__ cw, int MK, int Ds, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MK) return;
    float s = 0.f;
    for (int d = 0; d < Ds; ++d) s = __fmaf_rn(cw[(size_t) i * Ds + d], cw[(size_t) i * Ds + d], s);
    out[i] = s;
}
hipError_t launch_codeword_norms(const float *d_codewords, int M, int Ks, int Ds, float *d_cnorm, hipStream_t st)
{
    const int MK = M * Ks;
    hipLaunchKernelGGL(codeword_norms_kernel, dim3((MK + 255) / 256), dim3(256), 0, st, d_codewords, MK, Ds,
                       d_cnorm);
    return hipGetLastError();
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid: (ceil(Ks/16), M, ceil(B/16)); one wave per block computes a 16(query) x 16(codeword) tile of one m.
// A[i=lane&15][k=lane>>4] = q[b0+i][m*Ds + kk + k],  B[k=lane>>4][j=lane&15] = cw[m][ks0+j][kk + k]
// D: col = lane&15 (codeword), row = (lane>>4)*4 + r (query).
__global__ __launch_bounds__(64) void lut_build_mfma_kernel(const float *__restrict__ queries, int64_t B,
                                                            const float *__restrict__ codewords,
                                                            const float *__restrict__ cnorm, int M, int Ks,
                                                            int Ds, int QT, float *__restrict__ lut)
{
    const int lane = threadIdx.x;
    const int ks0 = blockIdx.x * 16, m = blockIdx.y;
    const int64_t b0 = (int64_t) blockIdx.z * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t bq = b0 + li;                 // query row this lane feeds into A
    const int ksb = ks0 + li;                   // codeword column this lane feeds into B
    const float *q = queries + (bq < B ? bq : 0) * (int64_t) (M * Ds) + (int64_t) m * Ds;
    const float *c = codewords + ((size_t) m * Ks + (ksb < Ks ? ksb : 0)) * Ds;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float qn = 0.f;                              // |q_m|^2 partial for row li over this lane's k slots
    for (int kk = 0; kk < Ds; kk += 4) {
        const int k = kk + lk;
        float a = (k < Ds && bq < B) ? q[k] : 0.f;
        float bv = (k < Ds && ksb < Ks) ? c[k] : 0.f;
        qn = __fmaf_rn(a, a, qn);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
    }
    // reduce |q|^2 over the 4 k-slots: lanes li, li+16, li+32, li+48
    qn += __shfl_xor(qn, 16);
    qn += __shfl_xor(qn, 32);
    const int MK = M * Ks;
    const int col = lane & 15;
    if (ks0 + col < Ks) {
        const float cn = cnorm[m * Ks + ks0 + col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) * 4 + r;
            const float qnr = __shfl(qn, row);   // lane `row` (< 16) holds |q_row|^2
            const int64_t b = b0 + row;
            if (b < B) {
                float v = qnr - 2.f * acc[r] + cn;
                lut[lut_index(b, m * Ks + ks0 + col, MK, QT)] = v < 0.f ? 0.f : v;
            }
        }
    }
}
hipError_t launch_lut_build_mfma(const float *d_queries, int64_t B, const float *d_codewords,
                                 const float *d_cnorm, int M, int Ks, int Ds, int QT, float *d_lut,
                                 hipStream_t st)
{
    if (B == 0) return hipSuccess;
    dim3 grid((Ks + 15) / 16, M, (unsigned) ((B + 15) / 16));
    hipLaunchKernelGGL(lut_build_mfma_kernel, grid, dim3(64), 0, st, d_queries, B, d_codewords, d_cnorm, M, Ks, Ds,
                       QT, d_lut);
    return hipGetLastError();
}

// ===================================================================================================
// (a3,a5) linear ADC scan + top-1 / key emission.
//   grid = (chunks, query tiles); block = 1024 threads = 16 waves; LDS = [M][Ks][QT] fp32 table.
//   lane = code, QT queries per lane; codes are read as 16-byte words; sum over m is sequential fp32.
// ===================================================================================================
template <int QT> struct LutT;
template <> struct LutT<1> { typedef float T; };
template <> struct LutT<2> { typedef float2 T; };
template <> struct LutT<4> { typedef float4 T; };

template <int QT> __device__ __forceinline__ void acc_add(float (&acc)[QT], const typename LutT<QT>::T &v);
template <> __device__ __forceinline__ void acc_add<1>(float (&acc)[1], const float &v)
{
    acc[0] = __fadd_rn(acc[0], v);
}
template <> __device__ __forceinline__ void acc_add<2>(float (&acc)[2], const float2 &v)
{
    acc[0] = __fadd_rn(acc[0], v.x);
    acc[1] = __fadd_rn(acc[1], v.y);
}
template <> __device__ __forceinline__ void acc_add<4>(float (&acc)[4], const float4 &v)
{
    acc[0] = __fadd_rn(acc[0], v.x);
    acc[1] = __fadd_rn(acc[1], v.y);
    acc[2] = __fadd_rn(acc[2], v.z);
    acc[3] = __fadd_rn(acc[3], v.w);
}

// ADC of one code whose M = 4*MW bytes are already in registers
template <int QT, int MW, int KST>
__device__ __forceinline__ void adc_words(const uint32_t (&w)[MW], const typename LutT<QT>::T *__restrict__ lut,
                                          float (&acc)[QT])
{
#pragma unroll
    for (int q = 0; q < QT; ++q) acc[q] = 0.f;
#pragma unroll
    for (int i = 0; i < MW; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t ks = (w[i] >> (8 * j)) & 0xffu;
            acc_add<QT>(acc, lut[(i * 4 + j) * KST + ks]);
        }
    }
}

// the same sums with the table entries of 16 subspaces at a time gathered BEFORE the first addition: sixteen LDS reads in flight instead
// of a read -> wait -> add chain per subspace (what the compiler makes of adc_words when it is short of nothing but patience)
template <int QT, int MW, int KST>
__device__ __forceinline__ void adc_words_gathered(const uint32_t (&w)[MW], const typename LutT<QT>::T *__restrict__ lut,
                                                   float (&acc)[QT])
{
    typedef typename LutT<QT>::T LT;
#pragma unroll
    for (int q = 0; q < QT; ++q) acc[q] = 0.f;
    constexpr int G = MW < 4 ? MW : 4;                // words per gather group (16 lookups)
#pragma unroll
    for (int i0 = 0; i0 < MW; i0 += G) {
        LT v[G * 4];
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i * 4 + j] = lut[((i0 + i) * 4 + j) * KST + ((w[i0 + i] >> (8 * j)) & 0xffu)];
#pragma unroll
        for (int t = 0; t < G * 4; ++t) acc_add<QT>(acc, v[t]);
    }
}

template <int MW> __device__ __forceinline__ void load_code_words(const uint8_t *__restrict__ p, uint32_t (&w)[MW])
{
    if constexpr (MW % 4 == 0) {
#pragma unroll
        for (int i = 0; i < MW / 4; ++i) {
            const uint4 v = reinterpret_cast<const uint4 *>(p)[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    } else if constexpr (MW % 2 == 0) {
#pragma unroll
        for (int i = 0; i < MW / 2; ++i) {
            const uint2 v = reinterpret_cast<const uint2 *>(p)[i];
            w[2 * i] = v.x; w[2 * i + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < MW; ++i) w[i] = reinterpret_cast<const uint32_t *>(p)[i];
    }
}

struct ScanArgs {
    const uint8_t *codes;
    int64_t n_codes;
    int M, Ks;
    const float *lut;
    int B;
    int tile0;
    int64_t chunk_len;
    unsigned long long *best;
    unsigned long long *keys;
    int b0;
    const int32_t *perm;           // scan position -> code id (scanorder.hip) or NULL when the codes are in id order
};

// rows per thread and trip of scan_kernel (> 1: the static-table form of the one / two-query top-1 instances, at most 64 KiB of table)
constexpr int scan_rows_per_trip(int QT, int MW, bool write_keys, bool perm)
{
    return (MW != 0 && !write_keys && !perm && QT <= 2 && QT * MW <= 16) ? (QT * MW <= 4 ? 8 : 4) : 1;
}

template <int QT, int MW, int KST, bool WRITE_KEYS, bool PERM = false>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(ScanArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename LutT<QT>::T LT;
    const int M = MW ? MW * 4 : p.M;
    const int Ks = KST ? KST : p.Ks;
    const int tid = threadIdx.x;
    const int tile = p.tile0 + blockIdx.y;
    const size_t lut_elems = (size_t) M * Ks * QT;
    // round 4: top-1 over one or two queries (4 / 8-byte table rows) is a pure stream on a large shard -- U rows per thread and trip are
    // requested before the first is scored, sixteen table reads of a row are in flight before the first addition (adc_words_gathered),
    // and the table sits in a STATIC LDS array: with its address known at compile time a lookup's address is one SDWA shift of the code
    // byte instead of a byte extract plus a shift-add onto the dynamic-LDS base (tools/probes/stream_probe.hip: 4.7 -> 5.3 TB/s).
    // A lane still meets its codes in ascending order, so the first minimum still wins.
    constexpr int U = scan_rows_per_trip(QT, MW, WRITE_KEYS, PERM);
    constexpr bool STATIC_TAB = U > 1;
    __shared__ typename LutT<QT>::T s_tab[STATIC_TAB ? (MW ? MW : 1) * 4 * 256 : 1];
    __shared__ unsigned long long s_red[QT];
    float *lds = STATIC_TAB ? reinterpret_cast<float *>(s_tab) : reinterpret_cast<float *>(smem);
    unsigned long long *red = STATIC_TAB ? s_red :
        reinterpret_cast<unsigned long long *>(smem + ((lut_elems * sizeof(float) + 15) & ~(size_t) 15));

    // ---- stage this tile's table: contiguous M*Ks*QT floats -> LDS (same layout) ----
    {
        const float *src = p.lut + (size_t) tile * lut_elems;
        if ((lut_elems & 3) == 0) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(lds);
            for (size_t i = tid; i < lut_elems / 4; i += kScanThreads) d4[i] = s4[i];
        } else {
            for (size_t i = tid; i < lut_elems; i += kScanThreads) lds[i] = src[i];
        }
        if (tid < QT) red[tid] = ~0ull;
    }
    __syncthreads();
    const LT *lut = reinterpret_cast<const LT *>(lds);

    const int64_t c_begin = (int64_t) blockIdx.x * p.chunk_len;
    int64_t c_end = c_begin + p.chunk_len;
    if (c_end > p.n_codes) c_end = p.n_codes;

    float bestd[QT];
    uint32_t besti[QT];
#pragma unroll
    for (int q = 0; q < QT; ++q) { bestd[q] = INFINITY; besti[q] = 0xffffffffu; }

    // round 4: one or two queries per block are HBM-latency-bound with one 16 / 32-byte row request per thread in flight (32 waves x 1 KiB
    // per CU against ~2 us of loaded latency = 4.2 TB/s, exactly what a 2 GB shard measured); four rows per thread are requested before
    // the first is scored.  A lane still meets its codes in ascending order, so the first minimum still wins.
    if constexpr (U > 1) {
        constexpr int MWc = MW ? MW : 1;
        int64_t n0 = c_begin + tid;
        const int64_t full_end = c_end - (int64_t) (U - 1) * kScanThreads;      // below it all U rows of a trip exist
        for (; n0 < full_end; n0 += (int64_t) kScanThreads * U) {
            uint32_t w[U][MWc];
            const uint8_t *row = p.codes + (size_t) n0 * (MWc * 4);
#pragma unroll
            for (int u = 0; u < U; ++u) load_code_words<MWc>(row + (size_t) u * kScanThreads * (MWc * 4), w[u]);
            __builtin_amdgcn_sched_barrier(0);         // all U requests leave before the first row is touched (the scheduler otherwise
                                                       // waits for row 0 with ONE request in flight and issues the others behind it)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float acc[QT];
                adc_words_gathered<QT, MWc, KST>(w[u], lut, acc);
#pragma unroll
                for (int q = 0; q < QT; ++q)
                    if (acc[q] < bestd[q]) { bestd[q] = acc[q]; besti[q] = (uint32_t) (n0 + (int64_t) u * kScanThreads); }
            }
        }
        for (; n0 < c_end; n0 += kScanThreads) {       // the last, partial trip
            uint32_t w[MWc];
            load_code_words<MWc>(p.codes + (size_t) n0 * (MWc * 4), w);
            float acc[QT];
            adc_words_gathered<QT, MWc, KST>(w, lut, acc);
#pragma unroll
            for (int q = 0; q < QT; ++q)
                if (acc[q] < bestd[q]) { bestd[q] = acc[q]; besti[q] = (uint32_t) n0; }
        }
    } else
    for (int64_t n = c_begin + tid; n < c_end; n += kScanThreads) {
        float acc[QT];
        if constexpr (MW != 0) {
            uint32_t w[MW ? MW : 1];
            load_code_words<MW>(p.codes + (size_t) n * (MW * 4), w);
            adc_words<QT, MW, KST>(w, lut, acc);
        } else {
            const uint8_t *c = p.codes + (size_t) n * M;
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[q] = 0.f;
            for (int m = 0; m < M; ++m) acc_add<QT>(acc, lut[m * Ks + c[m]]);
        }
        if constexpr (WRITE_KEYS) {
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const int b = tile * QT + q;
                if (b < p.B)
                    p.keys[(size_t) (b - p.b0) * p.n_codes + n] =
                        ((unsigned long long) f32_orderable(__float_as_uint(acc[q])) << 32) | (uint32_t) n;
            }
        } else {
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                // strict '<' + ascending n per lane => the first minimum wins, like a size-1 heap-select
                if (acc[q] < bestd[q]) { bestd[q] = acc[q]; besti[q] = (uint32_t) n; }
            }
            if constexpr (PERM) {
                // in the permuted scan order a lane's ids are not ascending: an exact tie (rare) goes to the smaller id
                bool tie = false;
#pragma unroll
                for (int q = 0; q < QT; ++q)          // besti == ~0: nothing chosen yet (acc = +inf meets the initial bestd)
                    tie |= (acc[q] == bestd[q]) & (besti[q] != (uint32_t) n) & (besti[q] != 0xffffffffu);
                if (__builtin_expect(tie, 0)) {
                    const int32_t idn = p.perm[n];
#pragma unroll
                    for (int q = 0; q < QT; ++q)
                        if (acc[q] == bestd[q] && besti[q] != (uint32_t) n && besti[q] != 0xffffffffu &&
                            idn < p.perm[besti[q]])
                            besti[q] = (uint32_t) n;
                }
            }
        }
    }
    if constexpr (PERM) {            // positions -> ids
#pragma unroll
        for (int q = 0; q < QT; ++q)
            if (besti[q] != 0xffffffffu) besti[q] = (uint32_t) p.perm[besti[q]];
    }

    if constexpr (!WRITE_KEYS) {
        // wave -> block -> global reduction of the packed key (orderable distance bits, then index):
        // min over the key == (dist asc, id asc), which for top-1 is exactly std::partial_sort's answer.
#pragma unroll
        for (int q = 0; q < QT; ++q) {
            unsigned long long key =
                besti[q] == 0xffffffffu
                    ? ~0ull
                    : (((unsigned long long) f32_orderable(__float_as_uint(bestd[q])) << 32) | besti[q]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(key, off);
                key = o < key ? o : key;
            }
            if ((tid & 63) == 0 && key != ~0ull) atomicMin(&red[q], key);
        }
        __syncthreads();
        if (tid < QT) {
            const int b = tile * QT + tid;
            if (b < p.B && red[tid] != ~0ull) atomicMin(&p.best[b], red[tid]);
        }
    }
}

template <int QT, int MW, int KST, bool WK, bool PERM = false>
static hipError_t launch_scan_t(const ScanParams &sp, int tile0, int ntiles, hipStream_t st)
{
    if constexpr (!WK && !PERM && QT == 4) {          // the scan-order variant exists for the 4-query tiles only
        if (sp.perm) return launch_scan_t<QT, MW, KST, WK, true>(sp, tile0, ntiles, st);
    }
    const size_t lut_bytes = (size_t) sp.M * sp.Ks * QT * sizeof(float);
    const size_t smem = scan_rows_per_trip(QT, MW, WK, PERM) > 1 ? 0 : ((lut_bytes + 15) & ~(size_t) 15) + 64;      // (static table)
    auto kern = scan_kernel<QT, MW, KST, WK, PERM>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    ScanArgs a;
    a.codes = sp.codes; a.n_codes = sp.n_codes; a.M = sp.M; a.Ks = sp.Ks; a.lut = sp.lut; a.B = sp.B;
    a.tile0 = tile0; a.chunk_len = sp.chunk_len; a.best = sp.best; a.keys = sp.keys; a.b0 = sp.b0;
    a.perm = PERM ? sp.perm : nullptr;
    launch_timed(kern, dim3(sp.chunks, ntiles), dim3(kScanThreads), smem, st, a);
    return hipGetLastError();
}

template <bool WK> static hipError_t launch_scan_wk(const ScanParams &sp, hipStream_t st)
{
    if (sp.n_codes == 0 || sp.B == 0) return hipSuccess;
    int tile0 = 0, ntiles = (sp.B + sp.QT - 1) / sp.QT;
    if (WK) { tile0 = sp.b0 / sp.QT; ntiles = (sp.b0 + sp.bc + sp.QT - 1) / sp.QT - tile0; }
    const bool fast = (sp.Ks == 256) && (sp.M % 4 == 0);
    if (fast && sp.QT == 4 && sp.M == 8) return launch_scan_t<4, 2, 256, WK>(sp, tile0, ntiles, st);
    if (fast && sp.QT == 4 && sp.M == 16) return launch_scan_t<4, 4, 256, WK>(sp, tile0, ntiles, st);
    if (fast && sp.QT == 4 && sp.M == 32) return launch_scan_t<4, 8, 256, WK>(sp, tile0, ntiles, st);
    if (fast && sp.QT == 2 && sp.M == 64) return launch_scan_t<2, 16, 256, WK>(sp, tile0, ntiles, st);
    if constexpr (!WK) {      // one or two queries: 4-byte table rows (ds_read_b32: 7 LDS cycles per 64 lookups instead of 11.7),
                              // which is what lets a single query stream codes at the HBM rate
        if (fast && sp.QT == 1 && sp.M == 8) return launch_scan_t<1, 2, 256, false>(sp, tile0, ntiles, st);
        if (fast && sp.QT == 1 && sp.M == 16) return launch_scan_t<1, 4, 256, false>(sp, tile0, ntiles, st);
        if (fast && sp.QT == 1 && sp.M == 32) return launch_scan_t<1, 8, 256, false>(sp, tile0, ntiles, st);
        if (fast && sp.QT == 1 && sp.M == 64) return launch_scan_t<1, 16, 256, false>(sp, tile0, ntiles, st);
        // two queries: 8-byte rows (ds_read_b64 costs what ds_read_b32 does; a 4-query tile's 16-byte rows make the pair LDS-bound)
        if (fast && sp.QT == 2 && sp.M == 8) return launch_scan_t<2, 2, 256, false>(sp, tile0, ntiles, st);
        if (fast && sp.QT == 2 && sp.M == 16) return launch_scan_t<2, 4, 256, false>(sp, tile0, ntiles, st);
        if (fast && sp.QT == 2 && sp.M == 32) return launch_scan_t<2, 8, 256, false>(sp, tile0, ntiles, st);
    }
    if (sp.QT == 4) return launch_scan_t<4, 0, 0, WK>(sp, tile0, ntiles, st);
    if (sp.QT == 2) return launch_scan_t<2, 0, 0, WK>(sp, tile0, ntiles, st);
    return launch_scan_t<1, 0, 0, WK>(sp, tile0, ntiles, st);
}

hipError_t launch_scan(const ScanParams &sp, hipStream_t st)
{
    return sp.keys ? launch_scan_wk<true>(sp, st) : launch_scan_wk<false>(sp, st);
}

__global__ void finalize_top1_kernel(const unsigned long long *__restrict__ best, int64_t B,
                                     const int64_t *__restrict__ remap, int64_t *__restrict__ out_ids,
                                     float *__restrict__ out_dists, int topk)
{
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long k = best[b];
    const uint32_t idx = (uint32_t) (k & 0xffffffffu);
    int64_t id = (k == ~0ull) ? -1 : (remap ? remap[idx] : (int64_t) idx);
    out_ids[b * topk] = id;
    out_dists[b * topk] = (k == ~0ull) ? INFINITY : __uint_as_float(f32_unorderable((uint32_t) (k >> 32)));
}
hipError_t launch_finalize_top1(const unsigned long long *d_best, int64_t B, const int64_t *d_remap,
                                int64_t *d_out_ids, float *d_out_dists, int topk, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(finalize_top1_kernel, dim3((unsigned) ((B + 255) / 256)), dim3(256), 0, st, d_best, B,
                       d_remap, d_out_ids, d_out_dists, topk);
    return hipGetLastError();
}

// rows of fully sorted packed keys -> first topk (ids, dists)
__global__ void gather_sorted_topk_kernel(const unsigned long long *__restrict__ sorted, int64_t bc,
                                          int64_t n_codes, int topk, const int64_t *__restrict__ remap,
                                          int64_t *__restrict__ out_ids, float *__restrict__ out_dists)
{
    const int64_t total = bc * topk;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t b = t / topk, k = t - b * topk;
        const unsigned long long key = sorted[b * n_codes + k];
        const uint32_t idx = (uint32_t) (key & 0xffffffffu);
        out_ids[t] = remap ? remap[idx] : (int64_t) idx;
        out_dists[t] = __uint_as_float(f32_unorderable((uint32_t) (key >> 32)));
    }
}
hipError_t launch_gather_sorted_topk(const unsigned long long *d_sorted, int64_t bc, int64_t n_codes, int topk,
                                     const int64_t *d_remap, int64_t *d_out_ids, float *d_out_dists,
                                     hipStream_t st)
{
    const int64_t total = bc * topk;
    if (total == 0) return hipSuccess;
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_sorted_topk_kernel, dim3(blocks), dim3(256), 0, st, d_sorted, bc, n_codes, topk,
                       d_remap, d_out_ids, d_out_dists);
    return hipGetLastError();
}

// sort path: flag the rows whose k+1 smallest distances hold an exact tie (the reference's order then depends on
// std::partial_sort's heap; tieorder.hip redoes those queries).  One thread per row pair.
__global__ void sorted_tie_flag_kernel(const unsigned long long *__restrict__ sorted, int64_t bc, int64_t n_codes, int topk,
                                       int32_t *__restrict__ flag_list, int *__restrict__ nflag)
{
    const int64_t k1 = (int64_t) topk + 1 < n_codes ? topk + 1 : n_codes;      // keys compared: rows [0, k1)
    const int64_t per = k1 - 1;
    const int64_t b = blockIdx.x;
    if (b >= bc || per <= 0) return;
    int tie = 0;
    for (int64_t j = threadIdx.x; j < per; j += blockDim.x)
        if ((sorted[b * n_codes + j] >> 32) == (sorted[b * n_codes + j + 1] >> 32)) tie = 1;
    if (__syncthreads_or(tie) && threadIdx.x == 0) flag_list[atomicAdd(nflag, 1)] = (int32_t) b;
}
hipError_t launch_sorted_tie_flag(const unsigned long long *d_sorted, int64_t bc, int64_t n_codes, int topk,
                                  int32_t *d_flag_list, int *d_nflag, hipStream_t st)
{
    if (bc == 0 || topk < 2) return hipSuccess;
    hipLaunchKernelGGL(sorted_tie_flag_kernel, dim3((unsigned) bc), dim3(256), 0, st, d_sorted, bc, n_codes, topk,
                       d_flag_list, d_nflag);
    return hipGetLastError();
}

// subset search: compact the target codes once per batch (src/rii.h:218-228 gathers per query)
__global__ void gather_codes_kernel(const uint8_t *__restrict__ codes, int M, const int64_t *__restrict__ ids,
                                    int64_t S, uint8_t *__restrict__ out)
{
    const int64_t total = S * M;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t s = t / M;
        const int m = (int) (t - s * M);
        out[t] = codes[(size_t) ids[s] * M + m];
    }
}
// round 4: the codes in POSTING order (entry p of the CSR id array -> row p): the candidates of a list become one contiguous run
__global__ void gather_codes_i32_kernel(const uint8_t *__restrict__ codes, int M, const int32_t *__restrict__ ids, int64_t S, uint8_t *__restrict__ out)
{
    const int64_t total = S * M;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t s2 = t / M;
        const int m = (int) (t - s2 * M);
        out[t] = codes[(size_t) ids[s2] * M + m];
    }
}
hipError_t launch_gather_codes_i32(const uint8_t *d_codes, int M, const int32_t *d_ids, int64_t S, uint8_t *d_out, hipStream_t st)
{
    if (S == 0) return hipSuccess;
    const int64_t total = S * M;
    hipLaunchKernelGGL(gather_codes_i32_kernel, dim3((unsigned) std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, st, d_codes, M, d_ids, S, d_out);
    return hipGetLastError();
}
hipError_t launch_gather_codes(const uint8_t *d_codes, int M, const int64_t *d_ids, int64_t S, uint8_t *d_out,
                               hipStream_t st)
{
    const int64_t total = S * M;
    if (total == 0) return hipSuccess;
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_codes_kernel, dim3(blocks), dim3(256), 0, st, d_codes, M, d_ids, S, d_out);
    return hipGetLastError();
}

// ===================================================================================================
// (a6) coarse scoring: one block per query, table of that query in LDS, lane = coarse centre.
// ===================================================================================================
__device__ __forceinline__ void stage_single_lut(const float *__restrict__ lut, int64_t b, int MK, int QT,
                                                 float *lds)
{
    const float *src = lut + (size_t) (b / QT) * MK * QT + (b % QT);
    for (int i = threadIdx.x; i < MK; i += blockDim.x) lds[i] = src[(size_t) i * QT];
}

__global__ __launch_bounds__(256) void ivf_coarse_kernel(IvfParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const int64_t bl = blockIdx.x;                   // local query index
    const int MK = p.M * p.Ks;
    stage_single_lut(p.lut, p.b0 + bl, MK, p.QT, lds);
    __syncthreads();
    for (int c = threadIdx.x; c < p.nlist; c += blockDim.x) {
        const uint8_t *code = p.centers + (size_t) c * p.M;
        float dist = 0.f;
        for (int m = 0; m < p.M; ++m) dist = __fadd_rn(dist, lds[m * p.Ks + code[m]]);
        p.coarse_dist[bl * p.nlist + c] = dist;
        p.coarse_id[bl * p.nlist + c] = c;
    }
}
hipError_t launch_ivf_coarse(const IvfParams &p, hipStream_t st)
{
    if (p.B == 0) return hipSuccess;
    const size_t smem = (size_t) p.M * p.Ks * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_coarse_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ivf_coarse_kernel, dim3((unsigned) p.B), dim3(256), smem, st, p);
    return hipGetLastError();
}

// ===================================================================================================
// (a6,a7) traversal plan: one lane per query: partial_sort the coarse scores (first w), then walk the
// lists in that order accumulating candidate counts until the reference's stop rule fires
// (src/rii.h:286-321): exactly L candidates, or >= topk after list number w, else "not found".
// ===================================================================================================
__global__ __launch_bounds__(64) void ivf_plan_kernel(IvfParams p)
{
    const int64_t bl = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (bl >= p.B) return;
    if (p.flag && !p.flag[bl]) return;          // already planned by ivf_fused_kernel
    int32_t *ids = p.coarse_id + bl * p.nlist;
    float *ds = p.coarse_dist + bl * p.nlist;
    pq_partial_sort(ids, ds, (long) p.w, (long) p.nlist);
    int32_t *cum = p.cum + bl * (int64_t) (p.nlist + 1);
    int64_t cnt = 0;
    int nv = 0;
    bool finished = false;
    for (int c = 0; c < p.nlist; ++c) {
        const int no = ids[c];
        const int64_t len = p.list_len[no];
        cum[c] = (int32_t) cnt;
        if (cnt + len >= p.L) { cnt = p.L; nv = c + 1; finished = true; break; }
        cnt += len;
        if ((int64_t) (c + 1) == p.w && cnt >= p.topk) { nv = c + 1; finished = true; break; }
    }
    if (!finished) { cnt = 0; nv = 0; }
    cum[nv] = (int32_t) cnt;
    p.ncand[bl] = (int32_t) cnt;
    p.nvis[bl] = nv;
}
hipError_t launch_ivf_plan(const IvfParams &p, hipStream_t st)
{
    if (p.B == 0) return hipSuccess;
    hipLaunchKernelGGL(ivf_plan_kernel, dim3((unsigned) ((p.B + 63) / 64)), dim3(64), 0, st, p);
    return hipGetLastError();
}

// ===================================================================================================
// (a7) candidate scan: one block per query; candidate p of the traversal is located by binary search in
// the per-query cumulative counts, its code gathered by id (one contiguous M-byte read), ADC'd against the
// LDS table.  topk == 1: block arg-min over (dist, traversal position) == the reference's answer.
// ===================================================================================================
__global__ __launch_bounds__(256) void ivf_scan_kernel(IvfParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const int64_t bl = blockIdx.x;
    const int MK = p.M * p.Ks;
    unsigned long long &red = *reinterpret_cast<unsigned long long *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));
    if (p.flag && !p.flag[bl]) return;          // already answered by ivf_fused_kernel
    const int ncand = p.ncand[bl];
    const int nv = p.nvis[bl];
    const bool top1 = (p.topk == 1);
    if (threadIdx.x == 0) red = ~0ull;
    if (ncand == 0) {
        if (threadIdx.x == 0) p.out_counts[bl] = 0;
        return;
    }
    stage_single_lut(p.lut, p.b0 + bl, MK, p.QT, lds);
    __syncthreads();
    const int32_t *cum = p.cum + bl * (int64_t) (p.nlist + 1);
    const int32_t *order = p.coarse_id + bl * p.nlist;
    float bestd = INFINITY;
    uint32_t bestp = 0xffffffffu;
    int32_t bestid = -1;
    for (int pos = threadIdx.x; pos < ncand; pos += blockDim.x) {
        int lo = 0, hi = nv;                       // largest j with cum[j] <= pos
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] <= pos) lo = mid; else hi = mid;
        }
        const int no = order[lo];
        const int32_t id = p.pl_ids[p.pl_off[no] + (pos - cum[lo])];
        const uint8_t *code = p.codes + (size_t) id * p.M;
        float dist = 0.f;
        if ((p.M & 3) == 0) {
            const uint32_t *cw = reinterpret_cast<const uint32_t *>(code);
            for (int i = 0; i < p.M / 4; ++i) {
                const uint32_t w = cw[i];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dist = __fadd_rn(dist, lds[(i * 4 + j) * p.Ks + ((w >> (8 * j)) & 0xffu)]);
            }
        } else {
            for (int m = 0; m < p.M; ++m) dist = __fadd_rn(dist, lds[m * p.Ks + code[m]]);
        }
        if (top1) {
            if (dist < bestd) { bestd = dist; bestp = (uint32_t) pos; bestid = id; }
        } else {
            p.cand_id[bl * p.cand_stride + pos] = id;
            p.cand_dist[bl * p.cand_stride + pos] = dist;
        }
    }
    if (top1) {
        unsigned long long key =
            bestp == 0xffffffffu ? ~0ull
                                 : (((unsigned long long) f32_orderable(__float_as_uint(bestd)) << 32) | bestp);
        const unsigned long long mine = key;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off);
            key = o < key ? o : key;
        }
        if ((threadIdx.x & 63) == 0 && key != ~0ull) atomicMin(&red, key);
        __syncthreads();
        if (mine != ~0ull && mine == red) {
            p.out_ids[bl] = bestid;
            p.out_dists[bl] = bestd;
            p.out_counts[bl] = 1;
        }
    }
}
hipError_t launch_ivf_scan(const IvfParams &p, hipStream_t st)
{
    if (p.B == 0) return hipSuccess;
    const size_t smem = (((size_t) p.M * p.Ks * sizeof(float) + 15) & ~(size_t) 15) + 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_scan_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ivf_scan_kernel, dim3((unsigned) p.B), dim3(256), smem, st, p);
    return hipGetLastError();
}

// ===================================================================================================
// Exact emulation of ONE query for shapes of any size (round 3: ivf_exact_big_kernel; round 4: also called by the block of
// ivf_fused_kernel that flagged its own query): the (distance, list) and (distance, position) sequences live in a global scratch
// slice, only the HEAP of each std::partial_sort -- its first `middle` entries: w lists, topk candidates -- is in LDS (s_head), and
// the library's __heap_select streams the rest from memory (rii_device.h: wh_partial_sort_split).  Same moves as the reference
// (src/rii.h:259-326), any nlist <= N and any L <= N; a heap deeper than the wave code covers (w or topk above kWhSplitMaxHeap) is
// walked by one lane over the global array.  `lds`: the query's exact table (LDS, or global memory for the wide shapes).
// Block-uniform; all 256 threads of the block call it.
// ===================================================================================================
__device__ __forceinline__ void ivf_exact_big_query(const IvfParams &p, int64_t bl, const float *lds, pq64_t *s_head, int32_t *s_misc,
                                                    unsigned char *mine, int tid)
{
    const int nlist = p.nlist, w = (int) p.w, k = p.topk;
    pq64_t *gco = reinterpret_cast<pq64_t *>(mine);                     // [nlist] (coarse distance, list), in the reference's order afterwards
    pq64_t *gcand = gco + nlist;                                        // [L]     (distance, traversal position)
    int32_t *gcum = reinterpret_cast<int32_t *>(gcand + p.L);           // [nlist + 1]
    int32_t *gcid = gcum + (nlist + 1);                                 // [L]     id of the candidate at a traversal position
    const bool w_lds = w <= kWhSplitMaxHeap, k_lds = k <= kWhSplitMaxHeap;          // heap in LDS, walked by a wave
    for (int c = tid; c < nlist; c += blockDim.x) {                                  // src/rii.h:262-264
        const pq64_t e = pq64_make(exact_adist(lds, p.centers + (size_t) c * p.M, p.M, p.Ks), (uint32_t) c);
        if (w_lds && c < w) s_head[c] = e; else gco[c] = e;
    }
    __syncthreads();
    if (w_lds) {
        if (tid < 64) wh_partial_sort_split(s_head, gco + w, w, nlist, tid);        // src/rii.h:279-280 (wave 0)
        __syncthreads();
        for (int c = tid; c < w; c += blockDim.x) gco[c] = s_head[c];                // the whole order in one array
    } else if (tid == 0) {
        pq64_partial_sort(gco, w, nlist);                                            // one lane, global memory: correct, slow
    }
    __syncthreads();
    if (tid == 0) {
        long long cnt = 0;
        int nv = 0;
        bool finished = false;
        for (int c = 0; c < nlist; ++c) {                                            // src/rii.h:286-321
            const long long len = p.list_len[pq64_id(gco[c])];
            gcum[c] = (int) cnt;
            if (cnt + len >= p.L) { cnt = p.L; nv = c + 1; finished = true; break; }
            cnt += len;
            if ((long long) (c + 1) == p.w && cnt >= p.topk) { nv = c + 1; finished = true; break; }
        }
        if (!finished) { cnt = 0; nv = 0; }
        gcum[nv] = (int) cnt;
        s_misc[0] = (int) cnt; s_misc[1] = nv;
    }
    __syncthreads();
    const int ncand = s_misc[0], nv = s_misc[1];
    if (ncand == 0) {
        if (tid == 0) p.out_counts[bl] = 0;                                          // src/rii.h:324-325
        return;
    }
    for (int pos = tid; pos < ncand; pos += blockDim.x) {
        int lo = 0, hi = nv;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (gcum[mid] <= pos) lo = mid; else hi = mid;
        }
        const int no = (int) pq64_id(gco[lo]);
        const int32_t id = p.pl_ids[p.pl_off[no] + (pos - gcum[lo])];
        const pq64_t e = pq64_make(exact_adist(lds, p.codes + (size_t) id * p.M, p.M, p.Ks), (uint32_t) pos);
        if (k_lds && pos < k) s_head[pos] = e; else gcand[pos] = e;
        gcid[pos] = id;
    }
    __syncthreads();
    if (k_lds) {
        if (tid < 64) wh_partial_sort_split(s_head, gcand + k, k, ncand, tid);      // src/rii.h:312-313 (wave 0)
    } else if (tid == 0) {
        pq64_partial_sort(gcand, k, ncand);
    }
    if (tid == 0) p.out_counts[bl] = k;
    __syncthreads();
    for (int j = tid; j < k; j += blockDim.x) {
        const pq64_t e = k_lds ? s_head[j] : gcand[j];
        p.out_ids[bl * k + j] = gcid[pq64_id(e)];
        p.out_dists[bl * k + j] = pq64_dist(e);
    }
}

// ===================================================================================================
// (a6+a7 fused) the common case in ONE launch per batch: block per query -- table staged once, coarse scores kept
// in LDS, the w nearest lists picked by w+1 rounds of block arg-min over (dist, list id), the stop rule applied to
// those w lists, candidates scanned.  Whenever the reference's answer could depend on std::partial_sort's internal
// order -- two of the w+1 smallest coarse distances exactly equal, or the walk has to continue into the unsorted
// tail past list w (fewer than topk hits so far) -- the block raises flag[b] and the exact emulation kernels
// (ivf_plan_kernel / ivf_scan_kernel, gated on the flag) take over for that query.
// ===================================================================================================
constexpr int kFusedMaxW = 32;
constexpr int kFusedMaxNlist = 4096;

// sequential fp32 ADC (RiiCpp::ADist, src/rii.h:375-394) of one code against a table in LDS
__device__ __forceinline__ float adc_lds(const float *lds, const uint8_t *code, int M, int Ks)
{
    float dist = 0.f;
    if ((M & 3) == 0) {
        const uint32_t *cw = reinterpret_cast<const uint32_t *>(code);
        for (int i = 0; i < M / 4; ++i) {
            const uint32_t wd = cw[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) dist = __fadd_rn(dist, lds[(i * 4 + j) * Ks + ((wd >> (8 * j)) & 0xffu)]);
        }
    } else {
        for (int m = 0; m < M; ++m) dist = __fadd_rn(dist, lds[m * Ks + code[m]]);
    }
    return dist;
}
// the same for a code already in registers as MQ (<= 4) 16-byte pieces
__device__ __forceinline__ float adc_lds_wide(const float *lds, const uint4 (&cv)[4], int MQ, int Ks)
{
    float dist = 0.f;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        if (qd < MQ) {
            const uint32_t wds[4] = {cv[qd].x, cv[qd].y, cv[qd].z, cv[qd].w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    dist = __fadd_rn(dist, lds[((qd * 4 + i) * 4 + j) * Ks + ((wds[i] >> (8 * j)) & 0xffu)]);
        }
    }
    return dist;
}

// Instantiated per (top-1 / top-k, Ds == 4 / generic): the all-in-one kernel was 63 KB of code -- the size of the instruction
// cache two CUs share -- of which a top-1, Ds = 4 query runs a fraction.
// GDIST: the coarse scores of the query live in global scratch (p.coarse_dist, L2-resident: 4 nlist bytes) instead of LDS -- the
// form for nlist above kFusedMaxNlist (the reference's default nlist = sqrt(N) is 11 k at a 125 M-code shard and 31.6 k at 1e9
// codes: rii/rii.py:143); the selection rounds stream them from there.  Same code otherwise.
template <bool TOP1, bool DS4, bool LSEL = false, bool GDIST = false>
__global__ __launch_bounds__(256) void ivf_fused_kernel(IvfParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int MK = p.M * p.Ks;
    float *lds = reinterpret_cast<float *>(smem);
    unsigned char *base = smem + (((size_t) MK * 4 + 15) & ~(size_t) 15);
    const int SC = p.sel_cap;                                                            // kFusedMaxW + 2, or pow2 >= nlist
    unsigned long long *s_sel = reinterpret_cast<unsigned long long *>(base);            // [SC]
    unsigned long long *s_red = s_sel + SC;                                              // [2]
    int *s_cum = reinterpret_cast<int *>(s_red + 2);                                     // [SC + 2]
    int *s_misc = s_cum + (SC + 2);                                                      // [4]: ncand, nv, flag
    int *s_len = s_misc + 4;                                                             // [SC + 2] lengths of the selected lists ...
    int *s_poff = s_len + (SC + 2);                                                      // [SC + 2] ... and their offsets, in visiting order
    unsigned long long *s_wsel = reinterpret_cast<unsigned long long *>(                  // [4][kFusedMaxW + 1] the waves' own picks
        smem + ((reinterpret_cast<unsigned char *>(s_poff + (SC + 2)) - smem + 7) & ~(size_t) 7));
    float *s_region = reinterpret_cast<float *>(s_wsel + 4 * (kFusedMaxW + 1));           // [nlist] coarse scores; LSEL: later the
    uint32_t *s_cd = reinterpret_cast<uint32_t *>(s_region);                             //   candidates' orderable distances [<= L]
    // top-k > 1: key buffer behind that region (LSEL: p.kcap keys of the final sort; else the streaming buffer), 16-byte aligned
    const int nreg = GDIST ? 0 : p.nlist;
    const int region = LSEL ? (nreg > (int) p.L ? nreg : (int) p.L) : nreg;
    unsigned long long *s_buf = reinterpret_cast<unsigned long long *>(
        smem + ((reinterpret_cast<unsigned char *>(s_region + region) - smem + 15) & ~(size_t) 15));
    const int64_t bl = blockIdx.x;
    float *s_dist = GDIST ? (p.coarse_dist + bl * (int64_t) p.nlist) : s_region;
    const int tid = threadIdx.x;
    const int nlist = p.nlist;
    const int w = (int) p.w;
    if (bl == 0 && tid == 0 && p.nflag_next) *p.nflag_next = 0;
    // every exit of the block is uniform: the rows of this query (or its fallback flag) are out -> tell a spinning host (host_spin)
    auto publish = [&]() {
        if (p.host_flag) {
            // every thread's row stores have left the CU before the barrier (coherent host memory is uncached on the device: a drained
            // store is on its way to the host ahead of the flag; the workgroup-scope barrier alone does not wait for vmcnt: ADVICE r3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __threadfence_system();
                __hip_atomic_store(&p.host_flag[p.b0 + bl], p.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    };

    if (p.queries) {
        // table built in place (exact fvec_L2sqr order): no global round trip for the common case
        const float *q = p.queries + (p.b0 + bl) * (int64_t) (p.M * p.Ds);
        if constexpr (DS4) {
            // 16 independent 16-byte codeword loads in flight per thread (the block has nothing else to hide them behind; this phase
            // is at the start of the kernel, where few other values are live: 8 in flight meant four dependent L2 round trips for
            // the M = 32 table, 16 mean two)
            const float4 *cw4 = reinterpret_cast<const float4 *>(p.codewords);
            const float4 *q4 = reinterpret_cast<const float4 *>(q);
            if (p.Ks == 256 && p.q_host_off) {
                // round 4: the query lives in the caller's pinned HOST block (a one-query call: no H2D copy in front of the launch).  Every
                // read of it crosses PCIe, so it is fetched ONCE: M threads load one sub-vector each, the load stays parked in a register
                // while the first codeword batch is requested, then goes to LDS; the table entries read it from there.
                float4 *s_q = reinterpret_cast<float4 *>(smem + p.q_host_off);
                float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tid < p.M) qv = q4[tid];
                for (int m0 = 0; m0 < p.M; m0 += 16) {
                    float4 cv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) cv[u] = (m0 + u < p.M) ? cw4[(m0 + u) * 256 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m0 == 0) {
                        if (tid < p.M) s_q[tid] = qv;
                        __syncthreads();
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (m0 + u < p.M) lds[(m0 + u) * 256 + tid] = fvec_l2sqr_ds4v(s_q[m0 + u], cv[u]);
                }
            } else if (p.Ks == 256) {
                // Ks = 256 = the block size: thread t owns entry ks = t of every subspace, so the subspace index -- and with it the
                // query's sub-vector -- is uniform across the block (scalar loads), and no entry needs an integer division by a
                // run-time Ks (which cost more vector instructions than the eleven of fvec_L2sqr itself: tools/ivf_phase_cost.py
                // measured 22 of the kernel's 38 us in this phase)
                for (int m0 = 0; m0 < p.M; m0 += 16) {
                    float4 cv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) cv[u] = (m0 + u < p.M) ? cw4[(m0 + u) * 256 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (m0 + u < p.M) lds[(m0 + u) * 256 + tid] = fvec_l2sqr_ds4v(q4[m0 + u], cv[u]);
                }
            } else
            for (int i0 = tid; i0 < MK; i0 += 256 * 16) {
                float4 cv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * 256;
                    cv[u] = i < MK ? cw4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * 256;
                    if (i < MK) lds[i] = fvec_l2sqr_ds4v(q4[i / p.Ks], cv[u]);
                }
            }
        } else if (p.Ks == 256 && p.Ds == 2) {
            // round 6: the reference's own harness setting is M = 64 over D = 128, i.e. Ds = 2 (examples/benchmark/ann_methods.py:19-34).
            // The plain loop below built that table in 72 us of the kernel's 180 (profiles/r06_fused_phases.json): one subspace per
            // dependent L2 round trip through the generic fvec_L2sqr.  Codewords of 16 subspaces in flight, arithmetic on registers.
            if (p.M == 64) table_rows_regs<2, 16, 64>(lds, q, p.codewords, p.M, p.arch, tid);
            else table_rows_regs<2, 16>(lds, q, p.codewords, p.M, p.arch, tid);
        } else if (p.Ks == 256 && p.Ds == 6) {
            table_rows_regs<6, 8>(lds, q, p.codewords, p.M, p.arch, tid);
        } else if (p.Ks == 256 && p.Ds == 8) {
            table_rows_regs<8, 4>(lds, q, p.codewords, p.M, p.arch, tid);
        } else
        for (int m = 0; m < p.M; ++m) {           // query sub-vector address is wave-uniform inside this loop
            const float *qm = q + (size_t) m * p.Ds;
            const float *cm = p.codewords + (size_t) m * p.Ks * p.Ds;
            for (int ks = tid; ks < p.Ks; ks += 256)
                lds[m * p.Ks + ks] = fvec_l2sqr_any(qm, cm + (size_t) ks * p.Ds, p.Ds, p.arch);
        }
    } else {
        stage_single_lut(p.lut, p.b0 + bl, MK, p.QT, lds);
    }
    __syncthreads();
    // every global load of this block is latency-exposed (one block per query, a few blocks per CU), so the gathers below
    // are issued in batches: up to four codes per thread, whole codes in registers before the first table lookup
    const bool wide = (p.M & 15) == 0 && p.M <= 64;
    const int rounds = (w + 1 < nlist) ? w + 1 : nlist;
    // round 4: up to 1024 lists and up to 33 picks (the common case): the keys of the four lists a thread scores stay in registers for the
    // selection (no LDS re-reads, no barrier between scoring and selection); the coarse scores still go to s_dist for the hand-over paths
    const bool kreg = nlist <= 4 * 256 && rounds <= kFusedMaxW + 1;
    unsigned long long kkey[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    for (int c0 = tid; c0 < nlist; c0 += 4 * 256) {
        if (wide) {
            uint4 cv[4][4];
            const int MQ = p.M >> 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 256;
                const uint4 *cp = reinterpret_cast<const uint4 *>(p.centers + (size_t) (c < nlist ? c : 0) * p.M);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    if (qd < MQ) cv[u][qd] = cp[qd];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 256;
                if (c < nlist) {
                    const float dv = adc_lds_wide(lds, cv[u], MQ, p.Ks);
                    s_dist[c] = dv;
                    kkey[u] = ((unsigned long long) f32_orderable(__float_as_uint(dv)) << 32) | (uint32_t) c;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 256;
                if (c < nlist) {
                    const float dv = adc_lds(lds, p.centers + (size_t) c * p.M, p.M, p.Ks);
                    s_dist[c] = dv;
                    kkey[u] = ((unsigned long long) f32_orderable(__float_as_uint(dv)) << 32) | (uint32_t) c;
                }
            }
        }
    }
    if (!kreg) __syncthreads();
    // ---- the w+1 smallest (dist, list id) keys, ascending: few -> two levels of wave-wide minima; many -> bitonic sort of all
    // list keys ----
    unsigned long long last = 0ull;
    if (rounds > kFusedMaxW + 1) {
        for (int c = tid; c < SC; c += blockDim.x)
            s_sel[c] = c < nlist ? (((unsigned long long) f32_orderable(__float_as_uint(s_dist[c])) << 32) | (uint32_t) c) : ~0ull;
        rr_bitonic_sort(s_sel, tid, SC);
    }
    bool staged = false;                  // wave 0 has already filled s_len / s_poff / s_cum / s_misc (fast form below)
    if (rounds <= kFusedMaxW + 1) {
        // Two levels, two barriers (round 2: w + 1 block-wide arg-min rounds with three barriers each, all of them exposed latency
        // in a block that has nothing else to run).  (a) Every wave extracts the `rounds` smallest keys among the lists its lanes
        // own, in ascending order, with wave-wide DPP minima (no LDS, no barrier) -- the w + 1 smallest keys of the block are among
        // the 4 x rounds found this way; (b) wave 0 extracts the `rounds` smallest of those the same way.
        const int wv = tid >> 6, ln = tid & 63;
        for (int r = 0; r < rounds; ++r) {
            unsigned long long best = ~0ull;
            if (kreg) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if ((r == 0 || kkey[u] > last) && kkey[u] < best) best = kkey[u];
            } else {
                for (int c = tid; c < nlist; c += blockDim.x) {
                    const unsigned long long key =
                        ((unsigned long long) f32_orderable(__float_as_uint(s_dist[c])) << 32) | (uint32_t) c;
                    if ((r == 0 || key > last) && key < best) best = key;
                }
            }
            last = wave_min_u64(best);                                         // ~0 when this wave's lists are exhausted
            if (ln == 0) s_wsel[wv * (kFusedMaxW + 1) + r] = last;
        }
        __syncthreads();
        if (wv == 0) {
            unsigned long long cand[3];                                        // 4 x 33 = 132 keys at most: up to three per lane
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = ln + 64 * u;                                     // i = wave * rounds + r
                cand[u] = i < 4 * rounds ? s_wsel[(i / rounds) * (kFusedMaxW + 1) + (i % rounds)] : ~0ull;
            }
            unsigned long long prev = 0ull, mysel = ~0ull;                     // lane r keeps pick r
            for (int r = 0; r < rounds; ++r) {
                unsigned long long best = ~0ull;
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if ((r == 0 || cand[u] > prev) && cand[u] < best) best = cand[u];
                prev = wave_min_u64(best);
                if (ln == r) mysel = prev;
            }
            // round 4: wave 0 goes straight on -- lane r fetches the length / offset of pick r, the stop rule of the walk (src/rii.h:
            // 283-326) is evaluated across the lanes (prefix sums of the lengths), and ONE barrier publishes everything; before: a
            // barrier, a block-wide fetch, a barrier, a one-lane loop over the picks, a barrier
            const int wl = w < nlist ? w : nlist;
            const uint32_t myhi = (uint32_t) (mysel >> 32);
            const uint32_t nxhi = (uint32_t) __shfl_down((int) myhi, 1);
            const bool tied = ln + 1 < rounds && myhi == nxhi;                  // exactly tied coarse distances among the w + 1 picks
            int len = 0, off = 0;
            if (ln < wl) {
                const int no = (int) (mysel & 0xffffffffu);
                len = p.list_len[no];
                off = (int) p.pl_off[no];                                       // N < 2^31
            }
            int incl = len;                                                     // inclusive prefix sums (the lists are disjoint: <= N)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (ln >= o) incl += t;
            }
            const int excl = incl - len;
            int flag = (p.force_flag || __ballot(tied) != 0ull) ? 1 : 0;
            const unsigned long long hit = __ballot(ln < wl && (long long) incl >= p.L);      // first list that completes L candidates
            int nv = 0;
            long long cnt = 0;
            if (hit) {
                nv = __ffsll((long long) hit);                                  // c1 + 1
                cnt = p.L;
            } else if ((long long) wl == p.w) {                                 // all w lists walked: enough for topk?
                const int tot = __shfl(incl, wl - 1);
                if (tot >= p.topk) { nv = wl; cnt = tot; } else flag = 1;
            } else flag = 1;                                                    // tail walk / empty return: exact path
            if (flag) { nv = 0; cnt = 0; }
            if (ln < wl) { s_len[ln] = len; s_poff[ln] = off; }
            if (ln < nv) s_cum[ln] = excl;
            if (ln == 0) {
                s_cum[nv] = (int) cnt;
                s_misc[0] = (int) cnt; s_misc[1] = nv; s_misc[2] = flag;
                p.flag[bl] = p.inl_scratch ? 0 : flag;
                if (flag && p.flag_list && !p.inl_scratch) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) bl;
                if (!flag) { p.ncand[bl] = (int) cnt; p.nvis[bl] = nv; }
                s_red[1] = ~0ull;
            }
            if (ln < rounds) s_sel[ln] = mysel;
        }
        staged = true;
    }
    if (!staged) {
        for (int c = tid; c < (w < nlist ? w : nlist); c += blockDim.x) {      // lengths / offsets of the lists the walk may visit
            const int no = (int) (s_sel[c] & 0xffffffffu);
            s_len[c] = p.list_len[no];
            s_poff[c] = (int) p.pl_off[no];                                    // N < 2^31
        }
        __syncthreads();
        if (tid == 0) {
            int flag = p.force_flag;
            for (int r = 0; r + 1 < rounds; ++r)
                if ((s_sel[r] >> 32) == (s_sel[r + 1] >> 32)) flag = 1;         // exactly tied coarse distances
            long long cnt = 0;
            int nv = 0;
            bool finished = false;
            const int wl = w < nlist ? w : nlist;
            for (int c = 0; c < wl && !flag; ++c) {
                const long long len = s_len[c];
                s_cum[c] = (int) cnt;
                if (cnt + len >= p.L) { cnt = p.L; nv = c + 1; finished = true; break; }
                cnt += len;
                if ((long long) (c + 1) == p.w && cnt >= p.topk) { nv = c + 1; finished = true; break; }
            }
            if (!finished) flag = 1;                                             // tail walk / empty return: exact path
            s_cum[nv] = (int) cnt;
            s_misc[0] = (int) cnt; s_misc[1] = nv; s_misc[2] = flag;
            p.flag[bl] = p.inl_scratch ? 0 : flag;
            if (flag && p.flag_list && !p.inl_scratch) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) bl;
            if (!flag) { p.ncand[bl] = (int) cnt; p.nvis[bl] = nv; }
            s_red[1] = ~0ull;
        }
    }
    __syncthreads();
    // round 4: the block that flagged its query redoes it ITSELF with the exact emulation (ivf_exact_big_query: sequences in this
    // query's global scratch slice, the heaps over the selection state in LDS -- dead by then -- and the table already in LDS) instead
    // of leaving it to a second, flag-gated launch that cost 4 - 6 us per batch with no query flagged
    auto redo_exact = [&]() {
        __syncthreads();                                   // every thread has read what it needed of the selection state
        pq64_t *s_head = reinterpret_cast<pq64_t *>(base);
        int32_t *xmisc = reinterpret_cast<int32_t *>(s_head + p.inl_hcap);
        if (tid == 0) { p.flag[bl] = 0; if (p.nflag) atomicAdd(p.nflag, 1); }       // (the counter: statistics only)
        ivf_exact_big_query(p, bl, lds, s_head, xmisc, p.inl_scratch + p.inl_per_q * (size_t) bl, tid);
        publish();
    };
    if (s_misc[2]) {
        if (p.inl_scratch) { redo_exact(); return; }
        // flagged: hand the coarse scores and the table (layout [b][M*Ks], QT == 1) to the exact-emulation kernels
        for (int c = tid; c < nlist; c += blockDim.x) {
            if constexpr (!GDIST) p.coarse_dist[bl * nlist + c] = s_dist[c];
            p.coarse_id[bl * nlist + c] = c;
        }
        if (p.queries) {
            float *dst = const_cast<float *>(p.lut) + (size_t) (p.b0 + bl) * MK;
            for (int i = tid; i < MK; i += blockDim.x) dst[i] = lds[i];
        }
        publish();
        return;
    }
    const int ncand = s_misc[0], nv = s_misc[1];
    constexpr bool top1 = TOP1;
    float bestd = INFINITY;
    uint32_t bestp = 0xffffffffu;
    int32_t bestid = -1;
    // round 4 (p.lcodes): the codes also exist in POSTING order (row pp = the code of posting pp of the CSR id array), so the candidates
    // of a list are one contiguous run: the code row is addressed from the traversal position alone -- no dependent id load in front of
    // it, coalesced 32-byte rows instead of random gathers -- and the id is fetched for the winner only
    for (int p0 = tid; top1 && p0 < ncand; p0 += 4 * 256) {
        int32_t id[4];                                // lcodes: the posting index; else the posting's id
#pragma unroll
        for (int u = 0; u < 4; ++u) {                 // up to four traversal positions per thread: loads in flight together
            const int pos = p0 + u * 256;
            id[u] = -1;
            if (pos < ncand) {
                int lo = 0, hi = nv;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_cum[mid] <= pos) lo = mid; else hi = mid;
                }
                const int pp = s_poff[lo] + (pos - s_cum[lo]);
                id[u] = p.lcodes ? pp : p.pl_ids[(size_t) pp];
            }
        }
        const uint8_t *cbase = p.lcodes ? p.lcodes : p.codes;
        float dist[4];
        if (wide) {
            uint4 cv[4][4];
            const int MQ = p.M >> 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4 *cp = reinterpret_cast<const uint4 *>(cbase + (size_t) (id[u] < 0 ? 0 : id[u]) * p.M);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    if (qd < MQ) cv[u][qd] = cp[qd];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) dist[u] = adc_lds_wide(lds, cv[u], MQ, p.Ks);
        } else {
            for (int u = 0; u < 4; ++u)
                dist[u] = id[u] < 0 ? INFINITY : adc_lds(lds, cbase + (size_t) id[u] * p.M, p.M, p.Ks);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)                   // ascending traversal position: strict < keeps the first minimum
            if (id[u] >= 0 && dist[u] < bestd) { bestd = dist[u]; bestp = (uint32_t) (p0 + u * 256); bestid = id[u]; }
    }
    if constexpr (TOP1) {
        unsigned long long key =
            bestp == 0xffffffffu ? ~0ull
                                 : (((unsigned long long) f32_orderable(__float_as_uint(bestd)) << 32) | bestp);
        const unsigned long long mine = key;
        key = wave_min_u64(key);                      // DPP minima (round 3's __shfl_xor ladder was twelve ds_bpermute round trips)
        if ((tid & 63) == 0 && key != ~0ull) atomicMin(&s_red[1], key);
        __syncthreads();
        if (mine != ~0ull && mine == s_red[1]) {
            // (the id next to every row instead -- four more loads per thread -- measured slower than this one dependent load)
            p.out_ids[bl] = p.lcodes ? p.pl_ids[(size_t) bestid] : bestid;
            p.out_dists[bl] = bestd;
            p.out_counts[bl] = 1;
        }
        publish();
        return;
    }
    if constexpr (!TOP1 && LSEL) {
        // ---- top-k > 1, L <= 4096: every candidate's distance goes to LDS (over the coarse scores, which are dead: a query
        // flagged from here on is redone by ivf_exact_lds_kernel, which computes its own), the (k+1)-th smallest is found by
        // bisection on the value bits and only the keys up to it are sorted.  A 64 .. 256-key buffer instead of the 2048-key
        // streaming buffer: four blocks per CU instead of three, so a 1024-query batch is one wave of blocks, not two. ----
        const int k1 = (p.topk + 1 < ncand) ? p.topk + 1 : ncand;
        const int kcap = p.kcap;
        __syncthreads();                                   // everybody is done with the coarse scores
        for (int p0 = tid; p0 < ncand; p0 += 4 * 256) {
            int32_t id[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pos = p0 + u * 256;
                id[u] = -1;
                if (pos < ncand) {
                    int lo = 0, hi = nv;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_cum[mid] <= pos) lo = mid; else hi = mid;
                    }
                    id[u] = p.lcodes ? (s_poff[lo] + (pos - s_cum[lo])) : p.pl_ids[(size_t) s_poff[lo] + (pos - s_cum[lo])];
                }
            }
            float dist[4];
            if (wide) {
                uint4 cv[4][4];
                const int MQ = p.M >> 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint4 *cp = reinterpret_cast<const uint4 *>((p.lcodes ? p.lcodes : p.codes) + (size_t) (id[u] < 0 ? 0 : id[u]) * p.M);
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        if (qd < MQ) cv[u][qd] = cp[qd];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) dist[u] = adc_lds_wide(lds, cv[u], MQ, p.Ks);
            } else {
                for (int u = 0; u < 4; ++u)
                    dist[u] = id[u] < 0 ? INFINITY : adc_lds(lds, (p.lcodes ? p.lcodes : p.codes) + (size_t) id[u] * p.M, p.M, p.Ks);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (id[u] >= 0) s_cd[p0 + u * 256] = f32_orderable(__float_as_uint(dist[u]));
        }
        unsigned int &s_cnt = *reinterpret_cast<unsigned int *>(&s_red[0]);
        // a bound on the k1-th smallest distance that leaves at most kcap keys under it: 256-bin histograms of the occupied range
        // (rii_device.h: block_kth_bound; round 2 bisected the 32 value bits with three barriers a bit).  The waves' list picks and
        // the selected keys are dead by now: their LDS serves as histogram and control words.
        __syncthreads();
        const uint32_t T = block_kth_bound([&](int i) { return s_cd[i]; }, ncand, (uint32_t) k1, (uint32_t) kcap,
                                           reinterpret_cast<unsigned int *>(s_wsel), reinterpret_cast<unsigned int *>(s_sel));
        __syncthreads();
        if (tid == 0) s_cnt = 0u;
        __syncthreads();
        for (int i0 = 0; i0 < ncand; i0 += 256) {          // keys up to the bound, appended wave by wave
            const int i = i0 + tid;
            const bool keep = i < ncand && s_cd[i] <= T;
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                unsigned int basepos = 0u;
                if ((tid & 63) == 0) basepos = atomicAdd(&s_cnt, (unsigned int) __popcll(bal));
                basepos = (unsigned int) __shfl((int) basepos, 0);
                const unsigned int at = basepos + (unsigned int) __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                if (keep && at < (unsigned int) kcap) s_buf[at] = ((unsigned long long) s_cd[i] << 32) | (uint32_t) i;
            }
        }
        __syncthreads();
        const unsigned int nkeep = s_cnt;
        int tie = nkeep > (unsigned int) kcap ? 1 : 0;     // more ties at the cut than the buffer holds: exact path
        if (!tie) {
            // up to 256 keys (= the block): every key counts the keys under it ((distance, position) pairs are distinct) and moves
            // to its rank -- two barriers instead of a bitonic ladder
            if (nkeep <= 256u) {
                const unsigned long long mine = tid < (int) nkeep ? s_buf[tid] : ~0ull;
                unsigned int rank = 0u;
                if (tid < (int) nkeep)
                    for (unsigned int j = 0; j < nkeep; ++j) rank += s_buf[j] < mine ? 1u : 0u;
                __syncthreads();
                if (tid < (int) nkeep) s_buf[rank] = mine;
                __syncthreads();
            } else {
                for (int i = tid; i < kcap; i += 256)
                    if ((unsigned int) i >= nkeep) s_buf[i] = ~0ull;
                rr_bitonic_sort(s_buf, tid, kcap);
            }
            for (int j = tid; j + 1 < k1; j += 256)
                if ((s_buf[j] >> 32) == (s_buf[j + 1] >> 32)) tie = 1;
        }
        if (__syncthreads_or(tie)) {                       // the answer hinges on std::partial_sort's heap order: hand over
            if (p.inl_scratch) { redo_exact(); return; }
            if (tid == 0) {
                p.flag[bl] = 1;
                if (p.flag_list) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) bl;
            }
            if (p.queries) {
                float *dst = const_cast<float *>(p.lut) + (size_t) (p.b0 + bl) * MK;
                for (int i = tid; i < MK; i += blockDim.x) dst[i] = lds[i];
            }
            publish();
            return;
        }
        for (int j = tid; j < p.topk; j += 256) {
            const unsigned long long key = s_buf[j];
            const int pos = (int) (key & 0xffffffffu);
            int lo = 0, hi = nv;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_cum[mid] <= pos) lo = mid; else hi = mid;
            }
            p.out_ids[bl * p.topk + j] = p.pl_ids[(size_t) s_poff[lo] + (pos - s_cum[lo])];
            p.out_dists[bl * p.topk + j] = __uint_as_float(f32_unorderable((uint32_t) (key >> 32)));
        }
        if (tid == 0) p.out_counts[bl] = p.topk;
        publish();
        return;
    }
    if constexpr (!TOP1 && !LSEL)
    // ---- top-k > 1: stream (dist, traversal position) keys through a block-local top-(k+1); if no two of those k+1
    // distances are equal the answer is independent of std::partial_sort's internals, else hand over to the emulation ----
    {
        const int k1 = (p.topk + 1 < ncand) ? p.topk + 1 : ncand;
        unsigned long long &s_kthr = s_buf[kRrBuf];
        unsigned int &s_cnt = *reinterpret_cast<unsigned int *>(&s_buf[kRrBuf + 1]);
        if (tid == 0) { s_cnt = 0u; s_kthr = ~0ull; }
        __syncthreads();
        for (int base = 0; base < ncand; base += 4 * 256) {       // four traversal positions per thread and trip (batched gathers)
            // the make-room decision must be uniform (the branch holds barriers): snapshot the counter between two barriers,
            // after every append of the previous trip and before any append of this one
            __syncthreads();
            const unsigned int cnt_now = s_cnt;
            __syncthreads();
            if (cnt_now + 4u * 256u > (unsigned int) kRrBuf) {
                for (int i = tid; i < kRrBuf; i += 256)
                    if ((unsigned int) i >= cnt_now) s_buf[i] = ~0ull;
                rr_bitonic_sort(s_buf, tid);
                if (tid == 0) { s_cnt = (unsigned int) k1; s_kthr = s_buf[k1 - 1]; }
                __syncthreads();
            }
            int32_t id[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pos = base + u * 256 + tid;
                id[u] = -1;
                if (pos < ncand) {
                    int lo = 0, hi = nv;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_cum[mid] <= pos) lo = mid; else hi = mid;
                    }
                    id[u] = p.lcodes ? (s_poff[lo] + (pos - s_cum[lo])) : p.pl_ids[(size_t) s_poff[lo] + (pos - s_cum[lo])];
                }
            }
            float dist[4];
            if (wide) {
                uint4 cv[4][4];
                const int MQ = p.M >> 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint4 *cp = reinterpret_cast<const uint4 *>((p.lcodes ? p.lcodes : p.codes) + (size_t) (id[u] < 0 ? 0 : id[u]) * p.M);
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        if (qd < MQ) cv[u][qd] = cp[qd];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) dist[u] = adc_lds_wide(lds, cv[u], MQ, p.Ks);
            } else {
                for (int u = 0; u < 4; ++u)
                    dist[u] = id[u] < 0 ? INFINITY : adc_lds(lds, (p.lcodes ? p.lcodes : p.codes) + (size_t) id[u] * p.M, p.M, p.Ks);
            }
            const unsigned long long thr = s_kthr;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (id[u] >= 0) {
                    const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(dist[u])) << 32) |
                                                   (uint32_t) (base + u * 256 + tid);
                    if (key < thr) s_buf[atomicAdd(&s_cnt, 1u)] = key;
                }
            }
        }
        __syncthreads();
        int nsort = 64;                           // smallest power of two covering the keys actually collected
        while (nsort < (int) s_cnt) nsort <<= 1;
        for (int i = tid; i < nsort; i += 256)
            if ((unsigned int) i >= s_cnt) s_buf[i] = ~0ull;
        rr_bitonic_sort(s_buf, tid, nsort);
        if (tid == 0) {
            int tie = 0;
            for (int j = 0; j + 1 < k1; ++j)
                if ((s_buf[j] >> 32) == (s_buf[j + 1] >> 32)) tie = 1;
            s_misc[2] = tie;
            if (tie && !p.inl_scratch) {
                p.flag[bl] = 1;
                if (p.flag_list) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) bl;
            }
        }
        __syncthreads();
        if (s_misc[2]) {          // ties at the cut, found only now: same hand-over as above
            if (p.inl_scratch) { redo_exact(); return; }
            for (int c = tid; c < nlist; c += blockDim.x) {
                if constexpr (!GDIST) p.coarse_dist[bl * nlist + c] = s_dist[c];
                p.coarse_id[bl * nlist + c] = c;
            }
            if (p.queries) {
                float *dst = const_cast<float *>(p.lut) + (size_t) (p.b0 + bl) * MK;
                for (int i = tid; i < MK; i += blockDim.x) dst[i] = lds[i];
            }
            publish();
            return;
        }
        for (int j = tid; j < p.topk; j += 256) {
            const unsigned long long key = s_buf[j];
            const int pos = (int) (key & 0xffffffffu);
            int lo = 0, hi = nv;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_cum[mid] <= pos) lo = mid; else hi = mid;
            }
            p.out_ids[bl * p.topk + j] = p.pl_ids[(size_t) s_poff[lo] + (pos - s_cum[lo])];
            p.out_dists[bl * p.topk + j] = __uint_as_float(f32_unorderable((uint32_t) (key >> 32)));
        }
        if (tid == 0) p.out_counts[bl] = p.topk;
        publish();
    }
}

// ===================================================================================================
// ivf_quad_kernel (round 5): FOUR queries per block of 1024 threads, top-1, Ds = 4, Ks = 256, M = 16 / 32, nlist <= 1024, w <= 32.
//
// ivf_fused_kernel keeps four one-query blocks on a CU; their coarse phase is 4 x nlist x M random 4-byte LDS reads (ds_read_b32:
// 32 banks, ~3.4 distinct rows on the busiest bank per 32-lane pass -- profiles/r04_ivf_pmc.json: 59 % of the LDS cycles of that
// kernel are bank conflicts) and every block reads the whole codebook and all the centres.  Here the four tables are interleaved
// [m][ks][query] in 16-byte rows:
//   tables   thread (ks, quarter of the subspaces): M / 4 codeword loads instead of M, the entry of all four queries from each
//            (the same eleven fp32 operations per entry, fvec_L2sqr's order), one 16-byte row written per entry -- conflict-free;
//   coarse   thread = centre: its code is read ONCE and every one of its M lookups is one ds_read_b128 that returns the entries of
//            all four queries (sequential fp32 adds over m per query, src/rii.h:375-384): a quarter of the LDS instructions, on
//            the 64-bank 16-byte path;
//   select   every wave extracts the w + 1 smallest keys of its 64 centres per query (DPP minima), then ONE WAVE PER QUERY merges
//            the sixteen waves' picks and evaluates the stop rule of the walk (src/rii.h:283-326) across its lanes -- the four
//            queries' selections run side by side;
//   scan     lane l of every wave works for query l mod 4 (the four queries' rows of a table entry lie in four different banks:
//            neighbouring lanes never collide), four candidates per thread in flight, first minimum in traversal order.
// A query whose answer could hinge on std::partial_sort's internals (exactly tied coarse distances among the w + 1 picks, a walk
// past list w) is flagged exactly as in ivf_fused_kernel and redone by the block afterwards with ivf_exact_big_query (its table
// de-interleaved through registers), or handed to the flag-gated exact kernels when no scratch slice was given.
// ===================================================================================================
constexpr int kQuadThreads = 1024;
constexpr int kQuadQ = 4;
constexpr int kQuadMaxNlist = 1024;
constexpr int kQuadR = kFusedMaxW + 1;             // picks per query (w + 1 <= 33)

__global__ __launch_bounds__(kQuadThreads) void ivf_quad_kernel(IvfParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = p.M, MK = M * 256;
    float4 *lds4 = reinterpret_cast<float4 *>(smem);                                     // [MK] rows of four queries' entries
    float *ldsf = reinterpret_cast<float *>(smem);
    unsigned char *base = smem + (size_t) MK * 16;
    unsigned long long *s_wsel = reinterpret_cast<unsigned long long *>(base);           // [4][16][kQuadR] the waves' picks
    unsigned long long *s_sel = s_wsel + kQuadQ * 16 * kQuadR;                           // [4][kQuadR + 1] the block's picks
    unsigned long long *s_red = s_sel + kQuadQ * (kQuadR + 1);                           // [4] best (distance, position)
    int *s_cum = reinterpret_cast<int *>(s_red + kQuadQ);                                // [4][kQuadR + 1]
    int *s_poff = s_cum + kQuadQ * (kQuadR + 1);                                         // [4][kQuadR + 1]
    int *s_misc = s_poff + kQuadQ * (kQuadR + 1);                                        // [4][4]: ncand, nv, flag
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nlist = p.nlist, w = (int) p.w;
    const int64_t q0 = (int64_t) blockIdx.x * kQuadQ;                                    // first query of the block (within the launch group)
    const int nq = (int) (p.B - q0 < kQuadQ ? p.B - q0 : kQuadQ);                        // live queries (the last block may hold fewer)
    if (blockIdx.x == 0 && tid == 0 && p.nflag_next) *p.nflag_next = 0;
    // the code of this thread's centre is requested before anything else: its L2 round trip passes behind the table phase instead of
    // in front of the coarse scores (-1.3 us of the block's dependent chain)
    const int MQ = M >> 4;                                                              // 16-byte pieces of a code
    uint4 cen[2];
    {
        const uint4 *cp = reinterpret_cast<const uint4 *>(p.centers + (size_t) (tid < nlist ? tid : 0) * M);
        cen[0] = cp[0];
        cen[1] = MQ > 1 ? cp[1] : make_uint4(0u, 0u, 0u, 0u);
    }

    // ---- tables: thread = (ks, quarter of the subspaces) ----
    {
        const int ks = tid & 255, mg = tid >> 8, mper = M >> 2;                          // (mg is wave-uniform: scalar query loads)
        const float4 *cw4 = reinterpret_cast<const float4 *>(p.codewords);
        const float4 *qv[kQuadQ];
#pragma unroll
        for (int q = 0; q < kQuadQ; ++q)
            qv[q] = reinterpret_cast<const float4 *>(p.queries + (p.b0 + q0 + (q < nq ? q : 0)) * (int64_t) (M * 4));
        float4 cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cv[u] = u < mper ? cw4[(mg * mper + u) * 256 + ks] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (u < mper) {
                const int m = mg * mper + u;
                float4 row;
                row.x = fvec_l2sqr_ds4v(qv[0][m], cv[u]);
                row.y = fvec_l2sqr_ds4v(qv[1][m], cv[u]);
                row.z = fvec_l2sqr_ds4v(qv[2][m], cv[u]);
                row.w = fvec_l2sqr_ds4v(qv[3][m], cv[u]);
                lds4[m * 256 + ks] = row;
            }
        }
    }
    __syncthreads();
    if (p.kcap == 1) return;                              // (measurement only -- option "ivf_dbg_stop": the phases' shares, tools/r5_ivf_phases.py)
    // ---- coarse scores: thread = centre, four queries per lookup ----
    const int rounds = (w + 1 < nlist) ? w + 1 : nlist;
    unsigned long long kkey[kQuadQ] = {~0ull, ~0ull, ~0ull, ~0ull};
    if (tid < nlist) {
        const uint4 (&cvv)[2] = cen;
        float acc[kQuadQ] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            if (qd < MQ) {
                const uint32_t wds[4] = {cvv[qd].x, cvv[qd].y, cvv[qd].z, cvv[qd].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 row = lds4[((qd * 4 + i) * 4 + j) * 256 + ((wds[i] >> (8 * j)) & 0xffu)];
                        acc[0] = __fadd_rn(acc[0], row.x);
                        acc[1] = __fadd_rn(acc[1], row.y);
                        acc[2] = __fadd_rn(acc[2], row.z);
                        acc[3] = __fadd_rn(acc[3], row.w);
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < kQuadQ; ++q) kkey[q] = ((unsigned long long) f32_orderable(__float_as_uint(acc[q])) << 32) | (uint32_t) tid;
    }
    // ---- level 1: the `rounds` smallest keys of this wave's 64 centres, per query, ascending (the four queries' DPP ladders in
    // lockstep: wave_min_u64_x4) ----
    // Round 6, `listed`: with at least `rounds` waves that hold centres, ONE ladder per wave gives the wave minima, the `rounds`-th
    // smallest of them bounds the block's `rounds` smallest keys from above (the minima at or below it are that many keys already), only
    // keys at or below the bound -- a handful -- are listed, and wave q orders query q's list: `rounds` lockstep ladders on sixteen
    // waves (1.7 us of the kernel at w = 4) become one, plus two short ones on four waves (shard_coarse_quad_kernel's selection).
    // More than 64 keys at or below the bound (masses of exactly tied distances): the query is flagged.
    const bool listed = rounds <= 16 && nlist > (rounds - 1) * 64;                      // (block-uniform)
    unsigned long long *s_list = s_wsel + kQuadQ * 16;                                  // [4][64], behind the wave minima [4][16]
    if (listed) {
        unsigned long long got[kQuadQ] = {kkey[0], kkey[1], kkey[2], kkey[3]};
        wave_min_u64_x4(got);
        if (lane < kQuadQ) s_wsel[lane * 16 + wave] = lane == 0 ? got[0] : lane == 1 ? got[1] : lane == 2 ? got[2] : got[3];
        if (tid < kQuadQ) s_misc[tid * 4 + 3] = 0;
        __syncthreads();
        if (wave < kQuadQ) {
            unsigned long long cand = lane < 16 ? s_wsel[wave * 16 + lane] : ~0ull, bound = ~0ull;
            for (int r = 0; r < rounds; ++r) {
                bound = wave_min_u64(cand);
                if (cand == bound) cand = ~0ull;
            }
            if (lane == 0) s_red[wave] = bound;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kQuadQ; ++q)
            if (kkey[q] != ~0ull && kkey[q] <= s_red[q]) {
                const int slot = atomicAdd(&s_misc[q * 4 + 3], 1);
                if (slot < 64) s_list[q * 64 + slot] = kkey[q];
            }
    } else {
        unsigned long long mykey[kQuadQ] = {kkey[0], kkey[1], kkey[2], kkey[3]};
        for (int r = 0; r < rounds; ++r) {
            unsigned long long got[kQuadQ] = {mykey[0], mykey[1], mykey[2], mykey[3]};
            wave_min_u64_x4(got);                                                       // ~0 when the wave's centres are exhausted
#pragma unroll
            for (int q = 0; q < kQuadQ; ++q) {
                if (mykey[q] == got[q]) mykey[q] = ~0ull;                               // (keys are distinct: one lane gives its key up)
                if (lane == 0) s_wsel[(q * 16 + wave) * kQuadR + r] = got[q];
            }
        }
    }
    __syncthreads();
    if (p.kcap == 2) return;
    // ---- level 2 + stop rule: wave q for query q ----
    if (wave < kQuadQ) {
        const int q = wave;
        const int ncnd = 16 * rounds;                                                   // <= 528 keys: up to nine per lane
        unsigned long long mysel = ~0ull;                                               // lane r keeps pick r
        bool over = false;
        if (listed) {
            const int n = s_misc[q * 4 + 3];
            over = n > 64;
            unsigned long long cand = lane < n && lane < 64 ? s_list[q * 64 + lane] : ~0ull;
            for (int r = 0; r < rounds; ++r) {
                const unsigned long long got = wave_min_u64(cand);
                if (cand == got) cand = ~0ull;
                if (lane == r) mysel = got;
            }
        }
        auto merge = [&](auto ns) {
            constexpr int NS = decltype(ns)::value;
            unsigned long long cand[NS];
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                const int i = lane + 64 * u;                                            // i = wave' * rounds + r
                cand[u] = i < ncnd ? s_wsel[(q * 16 + i / rounds) * kQuadR + (i % rounds)] : ~0ull;
            }
            for (int r = 0; r < rounds; ++r) {
                unsigned long long best = cand[0];
#pragma unroll
                for (int u = 1; u < NS; ++u) best = cand[u] < best ? cand[u] : best;
                const unsigned long long got = wave_min_u64(best);
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    if (cand[u] == got) cand[u] = ~0ull;
                if (lane == r) mysel = got;
            }
        };
        if (!listed) { if (ncnd <= 128) merge(std::integral_constant<int, 2>{}); else merge(std::integral_constant<int, 9>{}); }
        // the stop rule of the walk across the lanes (the wave-0 code of ivf_fused_kernel, per query)
        const int wl = w < nlist ? w : nlist;
        const uint32_t myhi = (uint32_t) (mysel >> 32);
        const uint32_t nxhi = (uint32_t) __shfl_down((int) myhi, 1);
        const bool tied = lane + 1 < rounds && myhi == nxhi;                             // exactly tied coarse distances among the w + 1 picks
        int len = 0, off = 0;
        if (lane < wl) {
            const int no = (int) (mysel & 0xffffffffu);
            len = p.list_len[no];
            off = (int) p.pl_off[no];                                                   // N < 2^31
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int excl = incl - len;
        int flag = (p.force_flag || over || __ballot(tied) != 0ull) ? 1 : 0;
        const unsigned long long hit = __ballot(lane < wl && (long long) incl >= p.L);    // first list that completes L candidates
        int nv = 0;
        long long cnt = 0;
        if (hit) {
            nv = __ffsll((long long) hit);
            cnt = p.L;
        } else if ((long long) wl == p.w) {                                             // all w lists walked: enough for topk?
            const int tot = __shfl(incl, wl - 1);
            if (tot >= p.topk) { nv = wl; cnt = tot; } else flag = 1;
        } else flag = 1;                                                                // tail walk / empty return: exact path
        if (flag) { nv = 0; cnt = 0; }
        if (q >= nq) { flag = 0; nv = 0; cnt = 0; }                                     // (padding query of the last block)
        if (lane < wl) s_poff[q * (kQuadR + 1) + lane] = off;
        if (lane < nv) s_cum[q * (kQuadR + 1) + lane] = excl;
        if (lane == 0) {
            s_cum[q * (kQuadR + 1) + nv] = (int) cnt;
            s_misc[q * 4 + 0] = (int) cnt; s_misc[q * 4 + 1] = nv; s_misc[q * 4 + 2] = flag;
            s_red[q] = ~0ull;
            if (q < nq) {
                const int64_t bl = q0 + q;
                p.flag[bl] = p.inl_scratch ? 0 : flag;
                if (flag && p.flag_list && !p.inl_scratch) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) bl;
                if (!flag) { p.ncand[bl] = (int) cnt; p.nvis[bl] = nv; }
            }
        }
    }
    __syncthreads();
    if (p.kcap == 3) return;
    // ---- candidates: lane l works for query l mod 4; thread j = tid / 4 of its query takes positions j, j + 256, ... ----
    {
        const int q = tid & 3, j = tid >> 2;
        const int ncand = s_misc[q * 4 + 0], nv = s_misc[q * 4 + 1];
        const int *cum = s_cum + q * (kQuadR + 1), *poff = s_poff + q * (kQuadR + 1);
        float bestd = INFINITY;
        uint32_t bestp = 0xffffffffu;
        int32_t bestid = -1;
        const uint8_t *cbase = p.lcodes ? p.lcodes : p.codes;
        for (int p0 = j; p0 < ncand; p0 += 4 * 256) {
            int32_t id[4];                                // lcodes: the posting index; else the posting's id
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pos = p0 + u * 256;
                id[u] = -1;
                if (pos < ncand) {
                    int lo = 0, hi = nv;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (cum[mid] <= pos) lo = mid; else hi = mid;
                    }
                    const int pp = poff[lo] + (pos - cum[lo]);
                    id[u] = p.lcodes ? pp : p.pl_ids[(size_t) pp];
                }
            }
            uint4 cvv[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4 *cp = reinterpret_cast<const uint4 *>(cbase + (size_t) (id[u] < 0 ? 0 : id[u]) * M);
                cvv[u][0] = cp[0];
                cvv[u][1] = MQ > 1 ? cp[1] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float dist = 0.f;
#pragma unroll
                for (int qd = 0; qd < 2; ++qd) {
                    if (qd < MQ) {
                        const uint32_t wds[4] = {cvv[u][qd].x, cvv[u][qd].y, cvv[u][qd].z, cvv[u][qd].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                dist = __fadd_rn(dist, ldsf[(((qd * 4 + i) * 4 + jj) * 256 + ((wds[i] >> (8 * jj)) & 0xffu)) * 4 + q]);
                    }
                }
                // ascending traversal position: strict < keeps the first minimum
                if (id[u] >= 0 && dist < bestd) { bestd = dist; bestp = (uint32_t) (p0 + u * 256); bestid = id[u]; }
            }
        }
        const unsigned long long mine = bestp == 0xffffffffu ? ~0ull : (((unsigned long long) f32_orderable(__float_as_uint(bestd)) << 32) | bestp);
        {                                                 // the wave's minimum per query (its lanes with l mod 4 == qq), in lockstep
            unsigned long long got[kQuadQ];
#pragma unroll
            for (int qq = 0; qq < kQuadQ; ++qq) got[qq] = q == qq ? mine : ~0ull;
            wave_min_u64_x4(got);
#pragma unroll
            for (int qq = 0; qq < kQuadQ; ++qq)
                if (lane == 0 && got[qq] != ~0ull) atomicMin(&s_red[qq], got[qq]);
        }
        __syncthreads();
        if (mine != ~0ull && mine == s_red[q]) {          // (distance, position) keys are distinct: one winner per query
            const int64_t bl = q0 + q;
            p.out_ids[bl] = p.lcodes ? p.pl_ids[(size_t) bestid] : bestid;
            p.out_dists[bl] = bestd;
            p.out_counts[bl] = 1;
        }
    }
    // ---- flagged queries (rare): this block redoes them exactly, or hands their tables to the flag-gated exact kernels ----
    const int f0 = s_misc[2], f1 = s_misc[6], f2 = s_misc[10], f3 = s_misc[14];
    if (!(f0 | f1 | f2 | f3)) return;
    {
        // the flagged queries' tables out of the interleaved rows, through registers (MK / 1024 entries per thread and query)
        float keep[kQuadQ][8];
        const int per = MK / kQuadThreads;                // 8 (M = 32) or 4 (M = 16)
#pragma unroll
        for (int q = 0; q < kQuadQ; ++q)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                keep[q][u] = (u < per && s_misc[q * 4 + 2]) ? ldsf[(size_t) (tid + kQuadThreads * u) * 4 + q] : 0.f;
        __syncthreads();                                  // every thread has read what it needs of the tables and the selection state
        for (int q = 0; q < kQuadQ; ++q) {
            const int fl = q == 0 ? f0 : q == 1 ? f1 : q == 2 ? f2 : f3;
            if (!fl) continue;
            const int64_t bl = q0 + q;
            if (p.inl_scratch) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < per) ldsf[tid + kQuadThreads * u] = q == 0 ? keep[0][u] : q == 1 ? keep[1][u] : q == 2 ? keep[2][u] : keep[3][u];
                pq64_t *s_head = reinterpret_cast<pq64_t *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));
                int32_t *xmisc = reinterpret_cast<int32_t *>(s_head + p.inl_hcap);
                __syncthreads();
                if (tid == 0) { p.flag[bl] = 0; if (p.nflag) atomicAdd(p.nflag, 1); }       // (the counter: statistics only)
                ivf_exact_big_query(p, bl, ldsf, s_head, xmisc, p.inl_scratch + p.inl_per_q * (size_t) bl, tid);
                __syncthreads();
            } else {
                float *dst = const_cast<float *>(p.lut) + (size_t) (p.b0 + bl) * MK;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < per) dst[tid + kQuadThreads * u] = q == 0 ? keep[0][u] : q == 1 ? keep[1][u] : q == 2 ? keep[2][u] : keep[3][u];
            }
        }
    }
}

static size_t ivf_quad_smem(int M) { return (size_t) M * 256 * 16 + (size_t) kQuadQ * 16 * kQuadR * 8 + (size_t) kQuadQ * (kQuadR + 1) * 8 + kQuadQ * 8 +
                                            2 * (size_t) kQuadQ * (kQuadR + 1) * 4 + kQuadQ * 4 * 4 + 64; }
bool ivf_quad_supported(int M, int Ks, int Ds, int nlist, int64_t w, int topk)
{
    return topk == 1 && Ds == 4 && Ks == 256 && (M == 16 || M == 32) && nlist <= kQuadMaxNlist && nlist >= 1 && w <= kFusedMaxW &&
           ivf_quad_smem(M) <= (size_t) 160 * 1024 - 512;
}
hipError_t launch_ivf_quad(const IvfParams &p0, hipStream_t st)
{
    if (p0.B == 0) return hipSuccess;
    IvfParams p = p0;                                    // (p.kcap: the caller's debug stop, 0 = none)
    size_t smem = ivf_quad_smem(p.M);
    if (p.inl_scratch) {          // the flagged queries' exact replay: plain table + heap over the (dead) interleaved rows
        p.inl_hcap = ivf_exact_big_heap_cap(p.w, p.topk);
        smem = std::max(smem, (((size_t) p.M * p.Ks * 4 + 15) & ~(size_t) 15) + (size_t) p.inl_hcap * 8 + 64);
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_quad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    launch_timed(ivf_quad_kernel, dim3((unsigned) ((p.B + kQuadQ - 1) / kQuadQ)), dim3(kQuadThreads), smem, st, p);
    return hipGetLastError();
}

// ===================================================================================================
// ivf_rot_kernel (round 6): the one-query inverted-index block with a CONFLICT-FREE table gather.
//
// ivf_fused_kernel looks a table entry up as lds[m * 256 + code[m]] with thread = candidate: the 32 lanes of a DS service group hit
// banks code % 32 at random, ~3.5 rows on the busiest bank -- 65 % of that kernel's LDS cycles were bank-conflict replays at the
// reference's own harness setting (M = 64, L = 5000: profiles/r05_refharness_pmc.json), 81 of its 129 us the candidate phase.
// Here (tools/ubench/gather_rot.hip measured the loop alone: 111 -> 64 us for 6016 rows at M = 64, bit-identical sums):
//   table     [ks][64 columns] floats, column c = subspace c mod M (M = 32: two copies, so the rotation below never wraps inside a
//             row), 64 KiB whatever M: the ADDRESS of an entry is { byte 1 = code byte, byte 0 = 4 x column } -- one v_perm_b32;
//   lanes     skewed in time: in round j lane l works on subspace m = (j - phi) mod M of ITS OWN row, phi = l mod 32, so the 32
//             lanes of a DS group read 32 different columns = 32 different banks whatever the code bytes are.  A row is still summed
//             by one lane in the order m = 0 .. M-1 (RiiCpp::ADist, src/rii.h:375-394): the distance bits are the reference's;
//   rows      come from tile copies (64 rows per tile, row r rotated by r mod 32 bytes: in round j every lane uses byte j of its
//             registers).  A lane is between two rows inside an iteration (tile k's row from round phi on, tile k-1's before): the
//             first 8 code dwords of the two rows are merged once per iteration with v_bfi_b32;
//   sums      two accumulators by row parity in one register pair: acc = (t, t) * sel_j + acc with the lane's (1,0) / (0,1) for that
//             round (one v_pk_fma_f32; exact: t * 1 + a, t * 0 + a).  Odd iterations read sel_j with its halves swapped.  At the end
//             of iteration k the other half holds the finished row of tile k-1 for EVERY lane: one compare, then it is zeroed.
//             Rounds >= 31 need no lane-dependent pair (every lane is on its new row): 31 register pairs.
// Work: coarse scores = tiles of the rotated centres (wave w takes tiles w, w + 4, ...); candidates = tiles of the visited lists in
// the rotated posting-order copy (every list starts on a tile boundary there), same distribution; one drain iteration per wave and
// phase.  Selection of the w + 1 nearest lists, the walk's stop rule and the flag protocol are ivf_fused_kernel's; a flagged query
// rebuilds the plain [m][ks] table in place and runs the exact replay (or hands the table to the flag-gated kernels).
// ===================================================================================================
typedef float rot_f2 __attribute__((ext_vector_type(2)));

template <int M> struct RotLane {
    static constexpr int NPH = M < 32 ? M : 32, NSEL = NPH - 1, LW = (NSEL + 3) / 4, MW = M / 4;
    uint32_t lowmask[LW];              // bytes of code dword d whose round 4 d + b is < phi: still the previous tile's row
    uint32_t laneoff;                  // M <= 32: 4 x the lane's column in round 0 (the immediate offset adds 4 j)
    uint32_t offq[M == 64 ? MW : 1];   // M = 64: 4 x the lane's column in the four rounds of code dword d, one byte each
    rot_f2 sel[NSEL];                  // round j < 31: (1, 0) = the lane is on its new row, (0, 1) = still on the previous one
    __device__ __forceinline__ void init(int lane)
    {
        const int phi = lane % NPH;
#pragma unroll
        for (int d = 0; d < LW; ++d) {
            uint32_t mk = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) mk |= (4 * d + b < phi) ? (0xffu << (8 * b)) : 0u;
            lowmask[d] = mk;
        }
        laneoff = (uint32_t) (((M - phi) % M) * 4);
        if (M == 16) laneoff += (lane & 16) ? 64u : 0u;          // (16 subspaces: the second half of a DS group takes the other 16 banks)
        if (M == 64) {
#pragma unroll
            for (int d = 0; d < MW; ++d) {
                uint32_t o = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) o |= (uint32_t) ((((4 * d + b - phi) % M + M) % M) * 4) << (8 * b);
                offq[M == 64 ? d : 0] = o;
            }
        }
#pragma unroll
        for (int j = 0; j < NSEL; ++j) sel[j] = phi <= j ? rot_f2{1.f, 0.f} : rot_f2{0.f, 1.f};
    }
};

// n_items rows of this lane, one per iteration: row_of(k) -> the lane's rotated row of item k (16-byte pieces); emit(k, sum) once
// item k's sum is complete (at the end of iteration k + 1).  Wave-uniform control flow, no barrier.  The table sits at LDS address 0.
// Two row buffers in alternation (even iterations sum the row in A and request the next one into B, odd ones the reverse): the buffer
// about to be overwritten is the PREVIOUS row, whose first 8 dwords the merge has just read -- no register copies between iterations.
template <int N> __device__ __forceinline__ uint32_t rot_dw(const uint4 (&r)[N], int d)
{
    return (d & 3) == 0 ? r[d >> 2].x : (d & 3) == 1 ? r[d >> 2].y : (d & 3) == 2 ? r[d >> 2].z : r[d >> 2].w;
}
template <int M, class RowOf, class Emit>
__device__ __forceinline__ void rot_gather(const RotLane<M> &rl, int n_items, RowOf row_of, Emit emit)
{
    constexpr int MW = M / 4, LW = RotLane<M>::LW, NSEL = RotLane<M>::NSEL, RB = 16, NB = M / RB;
    if (n_items <= 0) return;
    rot_f2 acc = {0.f, 0.f};
    const rot_f2 one_zero = {1.f, 0.f};
    uint4 bufA[MW / 4], bufB[MW / 4];
    {
        const uint4 *cp = row_of(0);
#pragma unroll
        for (int q = 0; q < MW / 4; ++q) { bufA[q] = cp[q]; bufB[q] = make_uint4(0u, 0u, 0u, 0u); }
    }
    auto step = [&](uint4 (&cur)[MW / 4], uint4 (&prev)[MW / 4], int s, auto odd_tag) {
        constexpr bool ODD = decltype(odd_tag)::value;
        uint32_t xl[LW];                                         // rounds < 31: the previous row's bytes before the lane's phase, this row's after
#pragma unroll
        for (int d = 0; d < LW; ++d) asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(xl[d]) : "v"(rl.lowmask[d]), "v"(rot_dw(prev, d)), "v"(rot_dw(cur, d)));
        if (s + 1 < n_items) {                                   // the next item's row: requested an iteration ahead, over the previous row
            const uint4 *cp = row_of(s + 1);
#pragma unroll
            for (int q = 0; q < MW / 4; ++q) prev[q] = cp[q];
        }
        auto loads = [&](int b, rot_f2 (&t)[RB / 2]) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int j = b * RB + u;
                const uint32_t xw = (j >> 2) < LW ? xl[(j >> 2) < LW ? (j >> 2) : 0] : rot_dw(cur, j >> 2);
                float v;
                if (M == 64) {
                    const uint32_t ps = 0x0c0c0000u | ((4u + (j & 3)) << 8) | (uint32_t) (j & 3);   // D.b0 = S1.b(j & 3), D.b1 = S0.b(j & 3)
                    const uint32_t a = __builtin_amdgcn_perm(xw, rl.offq[M == 64 ? (j >> 2) : 0], ps);
                    v = *reinterpret_cast<const __attribute__((address_space(3))) float *>(a);
                } else {
                    const uint32_t ps = 0x0c0c0000u | ((4u + (j & 3)) << 8);                         // D.b0 = S1.b0, D.b1 = S0.b(j & 3)
                    const uint32_t a = __builtin_amdgcn_perm(xw, rl.laneoff, ps);
                    v = *reinterpret_cast<const __attribute__((address_space(3))) float *>(a + 4 * j);
                }
                if (u & 1) t[u >> 1].y = v; else t[u >> 1].x = v;
            }
        };
        auto fmas = [&](int b, const rot_f2 (&t)[RB / 2]) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int j = b * RB + u;
                if (j >= NSEL) {
                    // from round 31 on every lane is on its new row: a plain add into that half of the pair (exactly what t * 1 + acc
                    // computes) -- the dependent chain of a row then runs on v_add_f32's latency for its second half, not v_pk_fma_f32's
                    const float tv = (u & 1) ? t[u >> 1].y : t[u >> 1].x;
                    if (!ODD) acc.x = __fadd_rn(acc.x, tv); else acc.y = __fadd_rn(acc.y, tv);
                    continue;
                }
                const rot_f2 sj = j < NSEL ? rl.sel[j < NSEL ? j : 0] : one_zero;
                // src0 = the round's entry for both halves (it sits in one half of a pair), src1 = the routing pair (odd iterations: swapped)
                if (!ODD && !(u & 1)) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(t[u >> 1]), "v"(sj));
                if (!ODD && (u & 1)) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(t[u >> 1]), "v"(sj));
                if (ODD && !(u & 1)) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "+v"(acc) : "v"(t[u >> 1]), "v"(sj));
                if (ODD && (u & 1)) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(t[u >> 1]), "v"(sj));
            }
        };
        rot_f2 t[2][RB / 2];                                     // the reads of batch b + 1 go out ahead of batch b's dependent fmas
        loads(0, t[0]);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) loads(b + 1, t[(b + 1) & 1]);
            fmas(b, t[b & 1]);
        }
        const float fin = ODD ? acc.x : acc.y;                   // the row of the previous item, complete in every lane
        if (s >= 1) emit(s - 1, fin);
        if (ODD) acc.x = 0.f; else acc.y = 0.f;
    };
    for (int s = 0; s <= n_items; s += 2) {
        step(bufA, bufB, s, std::false_type{});
        if (s + 1 <= n_items) step(bufB, bufA, s + 1, std::true_type{});
    }
}

// table in the rotated layout: thread = ks, row ks = [64 columns]; the entries of 16 (8) subspaces on registers, written as 16-byte
// pieces (a column-wise 4-byte store would put the 32 lanes of a DS group on one bank)
template <int M, int DS>
__device__ __forceinline__ void rot_table_rows(float *__restrict__ lds, const float *__restrict__ q, const float *__restrict__ codewords, int arch, int tid)
{
    constexpr int U = 16, COPIES = 64 / M, NV = DS == 4 ? 1 : DS / 2;
    static_assert(M % U == 0, "subspaces per batch");
    static_assert(DS <= 4, "one SIMD variant");
    // the codewords of batch b + 1 are requested before batch b is worked on (the block has nothing else to hide an L2 round trip behind)
    typedef typename std::conditional<DS == 4, float4, float2>::type V;
    V cv[2][U][NV];
    auto request = [&](int m0, V (&dst)[U][NV]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const V *src = reinterpret_cast<const V *>(codewords + ((size_t) (m0 + u) * 256 + tid) * DS);
#pragma unroll
            for (int i = 0; i < NV; ++i) dst[u][i] = src[i];
        }
    };
    request(0, cv[0]);
#pragma unroll
    for (int b = 0; b < M / U; ++b) {
        const int m0 = b * U;
        if (b + 1 < M / U) request(m0 + U, cv[(b + 1) & 1]);
        float ent[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (DS == 4) {
                ent[u] = fvec_l2sqr_ds4v(reinterpret_cast<const float4 *>(q)[m0 + u], cv[b & 1][u][0]);
            } else {
                float xq[DS], yc[DS];
#pragma unroll
                for (int i = 0; i < DS; ++i) xq[i] = q[(m0 + u) * DS + i];
#pragma unroll
                for (int i = 0; i < DS / 2; ++i) { yc[2 * i] = cv[b & 1][u][i].x; yc[2 * i + 1] = cv[b & 1][u][i].y; }
                ent[u] = fvec_l2sqr_body(xq, yc, DS, RII_SIMD_AVX512);         // (Ds <= 4: the three SIMD variants coincide -- table_rows_regs, rii_device.h)
            }
        }
        // 16-byte stores.  The rows of neighbouring lanes are 256 bytes apart -- the same banks -- so the 8 lanes of a DS store group
        // would all land on one bank quad (8-way: 56 of the 64 cycles of every store, a fifth of the kernel's LDS cycles:
        // profiles/r06_refharness_pmc_counters_table_store_conflicts.txt).  Store step g of lane l takes the quad (g + l) mod 4 of the
        // batch: four different quads per group, 2-way.
        static_assert(U == 16, "four quads per batch");
        const int r = tid & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a0 = ent[4 * (g & 3) + c], a1 = ent[4 * ((g + 1) & 3) + c], a2 = ent[4 * ((g + 2) & 3) + c], a3 = ent[4 * ((g + 3) & 3) + c];
                v[c] = r == 0 ? a0 : r == 1 ? a1 : r == 2 ? a2 : a3;
            }
            const int qd = (g + r) & 3;
#pragma unroll
            for (int c = 0; c < COPIES; ++c) *reinterpret_cast<float4 *>(lds + tid * 64 + c * M + m0 + 4 * qd) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int M>
__global__ __launch_bounds__(256) void ivf_rot_kernel(IvfParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SC = kFusedMaxW + 2, kTab = 256 * 64 * 4;
    float *lds = reinterpret_cast<float *>(smem);                                         // [256][64]: LDS address 0 (rot_gather)
    unsigned char *base = smem + kTab;
    unsigned long long *s_sel = reinterpret_cast<unsigned long long *>(base);            // [SC]
    unsigned long long *s_red = s_sel + SC;                                              // [2]
    unsigned long long *s_wsel = s_red + 2;                                              // [4][kFusedMaxW + 1] the waves' own picks
    int *s_cum = reinterpret_cast<int *>(s_wsel + 4 * (kFusedMaxW + 1));                 // [SC + 2] candidates before visited list i
    int *s_misc = s_cum + (SC + 2);                                                      // [4]: ncand, nv, flag
    int *s_len = s_misc + 4;                                                             // [SC + 2]
    int *s_poff = s_len + (SC + 2);                                                      // [SC + 2] first posting of the selected lists
    int *s_toff = s_poff + (SC + 2);                                                     // [SC + 2] ... their first tile in rlcodes
    int *s_tcum = s_toff + (SC + 2);                                                     // [SC + 2] tiles before visited list i
    const int64_t bl = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nlist = p.nlist, w = (int) p.w, MK = M * 256;
    if (bl == 0 && tid == 0 && p.nflag_next) *p.nflag_next = 0;
    if (reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char *) smem) != 0) __builtin_trap();
    const float *q = p.queries + (p.b0 + bl) * (int64_t) (M * p.Ds);
    if (p.kcap == 9) return;                               // (measurement only: launch cost alone)

    // ---- table (RiiCpp::DTable, src/rii.h:361-373; fvec_L2sqr's order) ----
    if (p.Ds == 4) rot_table_rows<M, 4>(lds, q, p.codewords, p.arch, tid);
    else rot_table_rows<M, 2>(lds, q, p.codewords, p.arch, tid);
    __syncthreads();
    if (p.kcap == 1) return;                               // (measurement only -- option "ivf_dbg_stop": the phases' shares, tools/r6_rot_ab.py)
    RotLane<M> rl;
    rl.init(lane);

    // ---- coarse scores (src/rii.h:259-264): tiles of the rotated centres; the keys of the (<= 4) lists a lane scores stay in registers ----
    const int rounds = (w + 1 < nlist) ? w + 1 : nlist;
    unsigned long long kkey[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    {
        const int ntc = (nlist + 63) >> 6;
        const int n_my = ntc > wave ? (ntc - wave + 3) >> 2 : 0;
        rot_gather<M>(rl, n_my,
                      [&](int k) { return reinterpret_cast<const uint4 *>(p.rcent + ((size_t) (wave + 4 * k) * 64 + lane) * M); },
                      [&](int k, float dv) {
                          const int c = (wave + 4 * k) * 64 + lane;
                          const unsigned long long key = c < nlist ? (((unsigned long long) f32_orderable(__float_as_uint(dv)) << 32) | (uint32_t) c) : ~0ull;
                          if (k == 0) kkey[0] = key; else if (k == 1) kkey[1] = key; else if (k == 2) kkey[2] = key; else kkey[3] = key;
                      });
    }
    if (p.kcap == 2) { if (kkey[0] == 1ull) p.out_ids[bl] = 0; return; }
    // ---- the w + 1 smallest (distance, list) keys, ascending: every wave extracts its own with DPP minima, wave 0 merges them and
    // evaluates the stop rule of the walk (src/rii.h:283-326) across its lanes -- ivf_fused_kernel's selection ----
    {
        unsigned long long last = 0ull;
        for (int r = 0; r < rounds; ++r) {
            unsigned long long best = ~0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((r == 0 || kkey[u] > last) && kkey[u] < best) best = kkey[u];
            last = wave_min_u64(best);
            if (lane == 0) s_wsel[wave * (kFusedMaxW + 1) + r] = last;
        }
    }
    __syncthreads();
    if (wave == 0) {
        unsigned long long cand[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = lane + 64 * u;
            cand[u] = i < 4 * rounds ? s_wsel[(i / rounds) * (kFusedMaxW + 1) + (i % rounds)] : ~0ull;
        }
        unsigned long long prev = 0ull, mysel = ~0ull;                     // lane r keeps pick r
        for (int r = 0; r < rounds; ++r) {
            unsigned long long best = ~0ull;
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if ((r == 0 || cand[u] > prev) && cand[u] < best) best = cand[u];
            prev = wave_min_u64(best);
            if (lane == r) mysel = prev;
        }
        const int wl = w < nlist ? w : nlist;
        const uint32_t myhi = (uint32_t) (mysel >> 32);
        const uint32_t nxhi = (uint32_t) __shfl_down((int) myhi, 1);
        const bool tied = lane + 1 < rounds && myhi == nxhi;               // exactly tied coarse distances among the w + 1 picks
        int len = 0, off = 0, toff = 0;
        if (lane < wl) {
            const int no = (int) (mysel & 0xffffffffu);
            len = p.list_len[no];
            off = (int) p.pl_off[no];                                       // N < 2^31
            toff = p.rl_toff[no];
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int excl = incl - len;
        int flag = (p.force_flag || __ballot(tied) != 0ull) ? 1 : 0;
        const unsigned long long hit = __ballot(lane < wl && (long long) incl >= p.L);        // first list that completes L candidates
        int nv = 0;
        long long cnt = 0;
        if (hit) {
            nv = __ffsll((long long) hit);
            cnt = p.L;
        } else if ((long long) wl == p.w) {                                 // all w lists walked: enough for topk?
            const int tot = __shfl(incl, wl - 1);
            if (tot >= p.topk) { nv = wl; cnt = tot; } else flag = 1;
        } else flag = 1;                                                    // tail walk / empty return: exact path
        if (flag) { nv = 0; cnt = 0; }
        // tiles of the visited lists (the last one cut at L)
        const int take = lane < nv ? ((lane == nv - 1) ? (int) cnt - excl : len) : 0;
        const int ntl = (take + 63) >> 6;
        int tincl = ntl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(tincl, o);
            if (lane >= o) tincl += t;
        }
        if (lane < wl) { s_len[lane] = len; s_poff[lane] = off; s_toff[lane] = toff; }
        if (lane < nv) { s_cum[lane] = excl; s_tcum[lane] = tincl - ntl; }
        if (lane == (nv > 0 ? nv - 1 : 0)) s_tcum[nv] = nv > 0 ? tincl : 0;
        if (lane == 0) {
            s_cum[nv] = (int) cnt;
            s_misc[0] = (int) cnt; s_misc[1] = nv; s_misc[2] = flag;
            p.flag[bl] = p.inl_scratch ? 0 : flag;
            if (flag && p.flag_list && !p.inl_scratch) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) bl;
            if (!flag) { p.ncand[bl] = (int) cnt; p.nvis[bl] = nv; }
            s_red[1] = ~0ull;
        }
    }
    __syncthreads();
    if (s_misc[2]) {
        // flagged (exactly tied coarse distances, a walk past list w, not found): the plain [m][ks] table in place, then the exact replay
        // by this block (ivf_exact_big_query: sequences in this query's global scratch slice, heaps behind the table) or the hand-over
        __syncthreads();
        if (p.Ds == 4) {
            const float4 *cw4 = reinterpret_cast<const float4 *>(p.codewords);
            const float4 *q4 = reinterpret_cast<const float4 *>(q);
            for (int m0 = 0; m0 < M; m0 += 16) {
                float4 cv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) cv[u] = cw4[(m0 + u) * 256 + tid];
#pragma unroll
                for (int u = 0; u < 16; ++u) lds[(m0 + u) * 256 + tid] = fvec_l2sqr_ds4v(q4[m0 + u], cv[u]);
            }
        } else {
            table_rows_regs<2, 16>(lds, q, p.codewords, M, p.arch, tid);
        }
        __syncthreads();
        if (p.inl_scratch) {
            pq64_t *s_head = reinterpret_cast<pq64_t *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));
            int32_t *xmisc = reinterpret_cast<int32_t *>(s_head + p.inl_hcap);
            if (tid == 0) { p.flag[bl] = 0; if (p.nflag) atomicAdd(p.nflag, 1); }       // (the counter: statistics only)
            ivf_exact_big_query(p, bl, lds, s_head, xmisc, p.inl_scratch + p.inl_per_q * (size_t) bl, tid);
        } else {
            float *dst = const_cast<float *>(p.lut) + (size_t) (p.b0 + bl) * MK;
            for (int i = tid; i < MK; i += 256) dst[i] = lds[i];
        }
        return;
    }
    if (p.kcap == 3) return;
    // ---- candidates (src/rii.h:283-305): the tiles of the visited lists, in traversal order; first minimum by (distance, position) ----
    const int nv = s_misc[1];
    float bestd = INFINITY;
    uint32_t bestp = 0xffffffffu;
    int32_t bestpp = -1;
    {
        const int T = s_tcum[nv];
        const int n_my = T > wave ? (T - wave + 3) >> 2 : 0;                // (launcher: at most 128 tiles per wave)
        // the wave's items described once, lane k (and k + 64) for item k: tile in rlcodes, first position / posting of the tile, rows
        // that count.  The iterations fetch them with v_readlane -- no LDS round trip, no search in front of a row request.
        int d_gt[2], d_p0[2], d_pp0[2], d_nr[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = wave + 4 * (lane + 64 * h);
            d_gt[h] = 0; d_p0[h] = 0; d_pp0[h] = 0; d_nr[h] = 0;
            if (g < T) {
                int i = 0;
                for (int c = 1; c < nv; ++c) i = s_tcum[c] <= g ? c : i;
                const int t = g - s_tcum[i], c0 = s_cum[i], left = s_cum[i + 1] - c0 - t * 64;
                d_gt[h] = s_toff[i] + t;
                d_p0[h] = c0 + t * 64;
                d_pp0[h] = s_poff[i] + t * 64;
                d_nr[h] = left < 64 ? left : 64;
            }
        }
        auto item = [&](const int (&d)[2], int k) { return k < 64 ? __builtin_amdgcn_readlane(d[0], k) : __builtin_amdgcn_readlane(d[1], k - 64); };
        rot_gather<M>(rl, n_my,
                      [&](int k) { return reinterpret_cast<const uint4 *>(p.rlcodes + ((size_t) item(d_gt, k) * 64 + lane) * M); },
                      [&](int k, float dv) {
                          if (lane < item(d_nr, k) && dv < bestd) { bestd = dv; bestp = (uint32_t) (item(d_p0, k) + lane); bestpp = item(d_pp0, k) + lane; }
                      });
    }
    unsigned long long key = bestp == 0xffffffffu ? ~0ull : (((unsigned long long) f32_orderable(__float_as_uint(bestd)) << 32) | bestp);
    const unsigned long long mine = key;
    key = wave_min_u64(key);
    if (lane == 0 && key != ~0ull) atomicMin(&s_red[1], key);
    __syncthreads();
    if (mine != ~0ull && mine == s_red[1]) {                                // (distance, position) keys are distinct: one winner
        p.out_ids[bl] = p.pl_ids[(size_t) bestpp];
        p.out_dists[bl] = bestd;
        p.out_counts[bl] = 1;
    }
}

static size_t ivf_rot_smem() { return (size_t) 256 * 64 * 4 + (size_t) (kFusedMaxW + 2 + 2 + 4 * (kFusedMaxW + 1)) * 8 + (size_t) (5 * (kFusedMaxW + 4) + 4) * 4 + 64; }
bool ivf_rot_supported(int M, int Ks, int Ds, int nlist, int64_t w, int topk)
{
    return topk == 1 && Ks == 256 && (M == 32 || M == 64) && (Ds == 2 || Ds == 4) && nlist >= 1 && nlist <= 1024 && w <= kFusedMaxW;
}
// (the kernel describes at most 128 tiles per wave in registers: L / 64 full tiles + one partial tile per visited list)
bool ivf_rot_fits(int64_t L, int64_t w) { return L / 64 + w + 1 <= 4 * 128; }
hipError_t launch_ivf_rot(const IvfParams &p0, hipStream_t st)
{
    if (p0.B == 0) return hipSuccess;
    if (!p0.rcent || !p0.rlcodes || !p0.rl_toff || !p0.queries) return hipErrorInvalidValue;
    IvfParams p = p0;
    size_t smem = ivf_rot_smem();
    if (p.inl_scratch) {
        p.inl_hcap = ivf_exact_big_heap_cap(p.w, p.topk);
        smem = std::max(smem, (((size_t) p.M * p.Ks * 4 + 15) & ~(size_t) 15) + (size_t) p.inl_hcap * 8 + 64);
        if (smem > (size_t) 160 * 1024) return hipErrorInvalidValue;
    }
    auto kern = p.M == 64 ? ivf_rot_kernel<64> : ivf_rot_kernel<32>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    launch_timed(kern, dim3((unsigned) p.B), dim3(256), smem, st, p);
    return hipGetLastError();
}

// ---- the rotated tile copies ----
// one thread per output dword: tile row tr = tile * 64 + r, rotation (r mod 64) mod min(M, 32)
__global__ void rot_rows_kernel(const uint8_t *__restrict__ src, int64_t n_rows, int M, uint8_t *__restrict__ dst, int64_t n_tiles)
{
    const int MW = M >> 2;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles * 64 * MW) return;
    const int64_t tr = i / MW;
    const int d = (int) (i - tr * MW);
    const int rot = (int) (tr & 63) % (M < 32 ? M : 32);
    uint32_t v = 0;
    if (tr < n_rows)
        for (int b = 0; b < 4; ++b) v |= (uint32_t) src[tr * M + ((4 * d + b - rot) % M + M) % M] << (8 * b);
    reinterpret_cast<uint32_t *>(dst)[i] = v;
}
__global__ void rot_lists_kernel(const uint8_t *__restrict__ lcodes, const int64_t *__restrict__ pl_off, const int32_t *__restrict__ rl_toff, int nlist, int M,
                                 uint8_t *__restrict__ dst, int64_t n_tiles)
{
    const int MW = M >> 2;
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles * 64 * MW) return;
    const int64_t tr = i / MW;
    const int d = (int) (i - tr * MW);
    const int tile = (int) (tr >> 6), r = (int) (tr & 63);
    int lo = 0, hi = nlist;                                                 // the list whose tiles contain `tile`: rl_toff[lo] <= tile < rl_toff[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rl_toff[mid] <= tile) lo = mid; else hi = mid;
    }
    const int64_t row = (int64_t) (tile - rl_toff[lo]) * 64 + r;
    const int rot = r % (M < 32 ? M : 32);
    uint32_t v = 0;
    if (row < pl_off[lo + 1] - pl_off[lo]) {
        const uint8_t *s = lcodes + (pl_off[lo] + row) * M;
        for (int b = 0; b < 4; ++b) v |= (uint32_t) s[((4 * d + b - rot) % M + M) % M] << (8 * b);
    }
    reinterpret_cast<uint32_t *>(dst)[i] = v;
}
hipError_t launch_rot_rows(const uint8_t *d_src, int64_t n_rows, int M, uint8_t *d_dst, int64_t n_tiles, hipStream_t st)
{
    const int64_t n = n_tiles * 64 * (M >> 2);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rot_rows_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, d_src, n_rows, M, d_dst, n_tiles);
    return hipGetLastError();
}
hipError_t launch_rot_lists(const uint8_t *d_lcodes, const int64_t *d_pl_off, const int32_t *d_rl_toff, int nlist, int M, uint8_t *d_dst,
                            int64_t n_tiles, hipStream_t st)
{
    const int64_t n = n_tiles * 64 * (M >> 2);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rot_lists_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, d_lcodes, d_pl_off, d_rl_toff, nlist, M, d_dst, n_tiles);
    return hipGetLastError();
}

// LSEL (selection in LDS): candidates' distances over the coarse scores + a small key buffer (see ivf_fused_kernel)
constexpr int kFusedSelMaxL = 4096;
static int ivf_fused_kcap(int topk)
{
    int c = 64;
    while (c < 2 * (topk + 1)) c <<= 1;
    return c;
}
static size_t ivf_fused_smem(int M, int Ks, int nlist, int sel_cap, int topk, int64_t L, bool lsel, bool gdist = false)
{
    const size_t head = (((size_t) M * Ks * sizeof(float) + 15) & ~(size_t) 15) + (size_t) sel_cap * 8 + 16 + (size_t) (sel_cap + 2) * 4 + 16 +
                        (size_t) (sel_cap + 2) * 8 + 8 + (size_t) 4 * (kFusedMaxW + 1) * 8;
    const int64_t nreg = gdist ? 0 : nlist;
    const size_t region = (size_t) (lsel ? std::max<int64_t>(nreg, L) : nreg) * 4 + 32;
    const size_t keys = topk > 1 ? (lsel ? (size_t) ivf_fused_kcap(topk) * 8 : (size_t) (kRrBuf + 2) * 8) : 0;
    return head + region + keys;
}
int ivf_fused_sel_cap(int nlist, int64_t w);
// coarse scores in LDS while they fit next to the table (nlist <= kFusedMaxNlist), else in global scratch -- that form needs the
// w + 1 selection rounds (w <= kFusedMaxW): the all-keys sort of larger w would not fit either
static bool ivf_fused_gdist(int M, int Ks, int nlist, int64_t w, int topk)
{
    return nlist > kFusedMaxNlist || ivf_fused_smem(M, Ks, nlist, ivf_fused_sel_cap(nlist, w), topk, 0, false) > (size_t) 160 * 1024;
}
static bool ivf_fused_lsel(int M, int Ks, int nlist, int64_t w, int topk, int64_t L)
{
    const bool gd = ivf_fused_gdist(M, Ks, nlist, w, topk);
    return topk > 1 && L <= kFusedSelMaxL && ivf_fused_kcap(topk) <= 2048 &&
           ivf_fused_smem(M, Ks, nlist, ivf_fused_sel_cap(nlist, w), topk, L, true, gd) <= (size_t) 160 * 1024;
}
bool ivf_fused_supported(int M, int Ks, int nlist, int64_t w, int topk)
{
    if (topk + 1 > kRrBuf / 2) return false;
    if (!ivf_fused_gdist(M, Ks, nlist, w, topk)) return true;
    return w <= kFusedMaxW && ivf_fused_smem(M, Ks, nlist, ivf_fused_sel_cap(nlist, w), topk, 0, false, true) <= (size_t) 160 * 1024;
}
int ivf_fused_sel_cap(int nlist, int64_t w)
{
    if (w <= kFusedMaxW) return kFusedMaxW + 2;
    int c = 64;
    while (c < nlist) c <<= 1;
    return c;
}

template <bool GD> static hipError_t launch_ivf_fused_t(const IvfParams &p, bool lsel, size_t smem, hipStream_t st)
{
    const bool top1 = p.topk == 1, ds4 = p.Ds == 4;
    auto kern = top1 ? (ds4 ? ivf_fused_kernel<true, true, false, GD> : ivf_fused_kernel<true, false, false, GD>)
                     : lsel ? (ds4 ? ivf_fused_kernel<false, true, true, GD> : ivf_fused_kernel<false, false, true, GD>)
                            : (ds4 ? ivf_fused_kernel<false, true, false, GD> : ivf_fused_kernel<false, false, false, GD>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int) smem);
    if (e != hipSuccess) return e;
    launch_timed(kern, dim3((unsigned) p.B), dim3(256), smem, st, p);
    return hipGetLastError();
}

hipError_t launch_ivf_fused(const IvfParams &p0, hipStream_t st)
{
    if (p0.B == 0) return hipSuccess;
    IvfParams p = p0;
    const bool gd = ivf_fused_gdist(p.M, p.Ks, p.nlist, p.w, p.topk);
    const bool lsel = ivf_fused_lsel(p.M, p.Ks, p.nlist, p.w, p.topk, p.L);
    p.kcap = lsel ? ivf_fused_kcap(p.topk) : 0;
    size_t smem = ivf_fused_smem(p.M, p.Ks, p.nlist, p.sel_cap, p.topk, p.L, lsel, gd);
    if (p.q_host_off) {           // the query staged in LDS behind everything else (see the kernel)
        p.q_host_off = (int) ((smem + 15) & ~(size_t) 15);
        smem = (size_t) p.q_host_off + (size_t) p.M * 16;
    }
    if (p.inl_scratch) {          // the flagged blocks' own exact replay: its heap lies over the selection state behind the table
        p.inl_hcap = ivf_exact_big_heap_cap(p.w, p.topk);
        smem = std::max(smem, (((size_t) p.M * p.Ks * 4 + 15) & ~(size_t) 15) + (size_t) p.inl_hcap * 8 + 64);
        if (smem > (size_t) 160 * 1024) return hipErrorInvalidValue;
    }
    return gd ? launch_ivf_fused_t<true>(p, lsel, smem, st) : launch_ivf_fused_t<false>(p, lsel, smem, st);
}

// final re-ranking (src/rii.h:312-319): std::partial_sort over the candidates in traversal order
__global__ __launch_bounds__(64) void ivf_select_kernel(IvfParams p)
{
    const int64_t bl = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (bl >= p.B) return;
    if (p.flag && !p.flag[bl]) return;          // already answered by ivf_fused_kernel
    const int n = p.ncand[bl];
    if (n == 0) { p.out_counts[bl] = 0; return; }
    int32_t *ids = p.cand_id + bl * p.cand_stride;
    float *ds = p.cand_dist + bl * p.cand_stride;
    pq_partial_sort(ids, ds, (long) p.topk, (long) n);
    for (int k = 0; k < p.topk; ++k) {
        p.out_ids[bl * p.topk + k] = ids[k];
        p.out_dists[bl * p.topk + k] = ds[k];
    }
    p.out_counts[bl] = p.topk;
}
hipError_t launch_ivf_select(const IvfParams &p, hipStream_t st)
{
    if (p.B == 0) return hipSuccess;
    hipLaunchKernelGGL(ivf_select_kernel, dim3((unsigned) ((p.B + 63) / 64)), dim3(64), 0, st, p);
    return hipGetLastError();
}

// ===================================================================================================
// Exact emulation for the flagged queries, entirely in LDS (block per query, early exit unless flag[b]):
// the coarse (list, dist) pairs and the candidate (id, dist) pairs live in LDS, ONE lane re-runs libstdc++'s
// std::partial_sort on them (64-cycle LDS accesses instead of dependent global loads: ~50 us instead of ~1 ms for a
// query), the candidate scan in between is parallel.  Covers nlist <= kExactLdsMax and L <= kExactLdsMax; larger shapes
// use the global-memory kernels above.
// ===================================================================================================
constexpr int kExactLdsMax = 4096;

__global__ __launch_bounds__(256) void ivf_exact_lds_kernel(IvfParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // a small persistent grid walks the compact list of flagged queries (the kernel needs ~100 KiB of LDS per block:
    // launching one block per query of the batch would serialise thousands of empty blocks behind that footprint)
    const int nflag = p.flag_list ? *p.nflag : (int) p.B;
    for (int fi = blockIdx.x; fi < nflag; fi += gridDim.x) {
    const int64_t bl = p.flag_list ? p.flag_list[fi] : fi;
    if (!p.flag_list && p.flag && !p.flag[bl]) continue;
    __syncthreads();                                   // previous query's LDS contents are dead from here on
    const int MK = p.M * p.Ks;
    const int nlist = p.nlist;
    const int tid = threadIdx.x;
    float *lds = reinterpret_cast<float *>(smem);
    unsigned char *base = smem + (((size_t) MK * 4 + 15) & ~(size_t) 15);
    pq64_t *s_coarse = reinterpret_cast<pq64_t *>(base);              // [nlist]   (coarse distance, list id), sorted in place
    int32_t *s_cum = reinterpret_cast<int32_t *>(s_coarse + nlist);   // [nlist+1] cumulative candidate counts
    int32_t *s_misc = s_cum + (nlist + 1);                            // [4]
    const int lcap = (int) (p.L < kExactLdsMax ? p.L : kExactLdsMax);
    pq64_t *s_cand = reinterpret_cast<pq64_t *>(smem + ((reinterpret_cast<unsigned char *>(s_misc + 4) - smem + 15) & ~(size_t) 15));   // [lcap]
    int32_t *s_cpos = reinterpret_cast<int32_t *>(s_cand + lcap);     // [lcap] ids of the candidates (the packed entry carries the position)

    stage_single_lut(p.lut, p.b0 + bl, MK, p.QT, lds);               // handed over by ivf_fused_kernel (or built before)
    __syncthreads();
    for (int c = tid; c < nlist; c += blockDim.x)
        s_coarse[c] = pq64_make(exact_adist(lds, p.centers + (size_t) c * p.M, p.M, p.Ks), (uint32_t) c);
    __syncthreads();
    if (tid < 64) wh_partial_sort(s_coarse, (int) p.w, nlist, tid);                 // src/rii.h:279-280 (wave 0)
    if (tid == 0) {
        long long cnt = 0;
        int nv = 0;
        bool finished = false;
        for (int c = 0; c < nlist; ++c) {                                           // src/rii.h:286-321
            const long long len = p.list_len[pq64_id(s_coarse[c])];
            s_cum[c] = (int) cnt;
            if (cnt + len >= p.L) { cnt = p.L; nv = c + 1; finished = true; break; }
            cnt += len;
            if ((long long) (c + 1) == p.w && cnt >= p.topk) { nv = c + 1; finished = true; break; }
        }
        if (!finished) { cnt = 0; nv = 0; }
        s_cum[nv] = (int) cnt;
        s_misc[0] = (int) cnt; s_misc[1] = nv;
    }
    __syncthreads();
    const int ncand = s_misc[0], nv = s_misc[1];
    if (ncand == 0) {
        if (tid == 0) p.out_counts[bl] = 0;                                          // src/rii.h:324-325
        continue;
    }
    for (int pos = tid; pos < ncand; pos += blockDim.x) {
        int lo = 0, hi = nv;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_cum[mid] <= pos) lo = mid; else hi = mid;
        }
        const int no = (int) pq64_id(s_coarse[lo]);
        const int32_t id = p.pl_ids[p.pl_off[no] + (pos - s_cum[lo])];
        s_cand[pos] = pq64_make(exact_adist(lds, p.codes + (size_t) id * p.M, p.M, p.Ks), (uint32_t) pos);
        s_cpos[pos] = id;
    }
    __syncthreads();
    if (tid < 64) wh_partial_sort(s_cand, p.topk, ncand, tid);                       // src/rii.h:312-313 (wave 0)
    if (tid == 0) p.out_counts[bl] = p.topk;
    __syncthreads();
    for (int j = tid; j < p.topk; j += blockDim.x) {
        const pq64_t e = s_cand[j];
        p.out_ids[bl * p.topk + j] = s_cpos[pq64_id(e)];
        p.out_dists[bl * p.topk + j] = pq64_dist(e);
    }
    }   // flagged-query loop
}

bool ivf_exact_lds_supported(int M, int Ks, int nlist, int64_t L)
{
    const size_t lcap = (size_t) (L < kExactLdsMax ? L : kExactLdsMax);
    const size_t need = (((size_t) M * Ks * 4 + 15) & ~(size_t) 15) + (size_t) nlist * 8 + (size_t) (nlist + 1) * 4 + 16 + lcap * 12 + 64;
    return nlist <= kExactLdsMax && L <= kExactLdsMax && need <= 160 * 1024;
}

hipError_t launch_ivf_exact_lds(const IvfParams &p, hipStream_t st)
{
    if (p.B == 0) return hipSuccess;
    const size_t lcap = (size_t) (p.L < kExactLdsMax ? p.L : kExactLdsMax);
    const size_t smem = (((size_t) p.M * p.Ks * 4 + 15) & ~(size_t) 15) + (size_t) p.nlist * 8 + (size_t) (p.nlist + 1) * 4 + 16 +
                        lcap * 12 + 64;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ivf_exact_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    const unsigned grid = p.flag_list ? (unsigned) std::min<int64_t>(p.B, 256) : (unsigned) p.B;
    hipLaunchKernelGGL(ivf_exact_lds_kernel, dim3(grid), dim3(256), smem, st, p);
    return hipGetLastError();
}

// ===================================================================================================
// ivf_exact_big_kernel: ivf_exact_big_query (above ivf_fused_kernel) for the flagged queries of a batch, or for every query of a
// shape the fused kernel does not cover -- a persistent grid, one global scratch slice per block.
// ===================================================================================================
// GTAB: the query's table does not fit LDS at all (M * Ks * 4 B > 144 KiB, widetab.hip): it is read from global memory (L2).
template <bool GTAB>
__global__ __launch_bounds__(256) void ivf_exact_big_kernel(IvfParams p, unsigned char *scratch, size_t per_block, int hcap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nflag = p.flag_list ? *p.nflag : (int) p.B;
    const int MK = p.M * p.Ks, tid = threadIdx.x;
    float *lds_tab = reinterpret_cast<float *>(smem);
    pq64_t *s_head = reinterpret_cast<pq64_t *>(smem + (GTAB ? 0 : (((size_t) MK * 4 + 15) & ~(size_t) 15)));   // [hcap] the heap of the running sort
    int32_t *s_misc = reinterpret_cast<int32_t *>(s_head + hcap);                                      // [4]
    unsigned char *mine = scratch + per_block * blockIdx.x;
    for (int fi = blockIdx.x; fi < nflag; fi += gridDim.x) {
        const int64_t bl = p.flag_list ? p.flag_list[fi] : fi;
        if (!p.flag_list && p.flag && !p.flag[bl]) continue;
        __syncthreads();                                   // the previous query's LDS contents are dead from here on
        if constexpr (!GTAB) stage_single_lut(p.lut, p.b0 + bl, MK, p.QT, lds_tab);
        const float *lds = GTAB ? p.lut + (size_t) (p.b0 + bl) * MK : lds_tab;        // (GTAB: plain [b][M*Ks] tables, QT == 1)
        __syncthreads();
        ivf_exact_big_query(p, bl, lds, s_head, s_misc, mine, tid);
    }
}

static int ivf_exact_big_hcap(int64_t w, int topk)
{
    const int64_t a = w <= kWhSplitMaxHeap ? w : 0, b = topk <= kWhSplitMaxHeap ? topk : 0;
    return (int) std::max<int64_t>(std::max<int64_t>(a, b), 1);
}
int ivf_exact_big_heap_cap(int64_t w, int topk) { return ivf_exact_big_hcap(w, topk); }
bool ivf_exact_big_supported(int M, int Ks, int64_t w, int topk)
{
    if (lut_tile_for(M, Ks) == 0) return true;             // table read from global memory: only the heaps use LDS
    return (((size_t) M * Ks * 4 + 15) & ~(size_t) 15) + (size_t) ivf_exact_big_hcap(w, topk) * 8 + 64 <= (size_t) 160 * 1024;
}
size_t ivf_exact_big_scratch(int nlist, int64_t L) { return ((size_t) nlist * 12 + (size_t) L * 12 + 4 + 63) / 64 * 64; }
hipError_t launch_ivf_exact_big(const IvfParams &p, void *d_scratch, int grid, hipStream_t st)
{
    if (p.B == 0) return hipSuccess;
    const int hcap = ivf_exact_big_hcap(p.w, p.topk);
    const bool gtab = lut_tile_for(p.M, p.Ks) == 0;
    const size_t smem = (gtab ? 0 : (((size_t) p.M * p.Ks * 4 + 15) & ~(size_t) 15)) + (size_t) hcap * 8 + 64;
    auto kern = gtab ? ivf_exact_big_kernel<true> : ivf_exact_big_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned) grid), dim3(256), smem, st, p, static_cast<unsigned char *>(d_scratch),
                       ivf_exact_big_scratch(p.nlist, p.L), hcap);
    return hipGetLastError();
}

// ---- target-id filtering (src/rii.h:294-296 does a binary search per posting; here: one bitmap per batch
// and an order-preserving compaction of every list, so the traversal above is oblivious to S) ----
__global__ void bitmap_set_kernel(const int64_t *__restrict__ tids, int64_t S, uint32_t *__restrict__ bitmap)
{
    const int64_t s = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int64_t t = tids[s];
    atomicOr(&bitmap[t >> 5], 1u << (t & 31));
}
hipError_t launch_bitmap_set(const int64_t *d_tids, int64_t S, uint32_t *d_bitmap, hipStream_t st)
{
    if (S == 0) return hipSuccess;
    hipLaunchKernelGGL(bitmap_set_kernel, dim3((unsigned) ((S + 255) / 256)), dim3(256), 0, st, d_tids, S,
                       d_bitmap);
    return hipGetLastError();
}

// round 4: a pass covers 1024 postings (the mean list of config 3 / 4 is 977) -- the four id loads of a thread, then its four bitmap
// words, are in flight together, and ONE exclusive scan over the 16 (quarter, wave) ballot counts places every kept id: three barriers per
// pass instead of three per 256 postings behind two dependent loads each (subset inverted index: this kernel was ~15 us of a 67 us step)
__global__ __launch_bounds__(256) void filter_lists_kernel(const int64_t *__restrict__ pl_off,
                                                           const int32_t *__restrict__ pl_ids,
                                                           const uint32_t *__restrict__ bitmap,
                                                           int32_t *__restrict__ fids, int32_t *__restrict__ flen)
{
    __shared__ int s_cnt[16], s_off[17];
    __shared__ int base;
    const int no = blockIdx.x;
    const int64_t beg = pl_off[no], end = pl_off[no + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    for (int64_t i = beg; i < end; i += 1024) {
        int32_t id[4];
        uint32_t word[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                       // posting i + 256 u + tid: list order = (u, wave, lane) order
            const int64_t idx = i + 256 * u + threadIdx.x;
            id[u] = idx < end ? pl_ids[idx] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) word[u] = id[u] >= 0 ? bitmap[id[u] >> 5] : 0u;
        unsigned long long bal[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool keep = id[u] >= 0 && ((word[u] >> (id[u] & 31)) & 1u);
            bal[u] = __ballot(keep);
            if (lane == 0) s_cnt[4 * u + wave] = __popcll(bal[u]);
        }
        __syncthreads();                                    // (also orders `base` of the previous pass before this pass's reads)
        if (threadIdx.x < 64) {
            const int c = lane < 16 ? s_cnt[lane] : 0;
            int incl = c;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            if (lane < 16) s_off[lane] = incl - c;
            if (lane == 15) s_off[16] = incl;
        }
        __syncthreads();
        const int b0 = base;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((bal[u] >> lane) & 1ull) fids[beg + b0 + s_off[4 * u + wave] + __popcll(bal[u] & ((1ull << lane) - 1ull))] = id[u];
        __syncthreads();                                    // everybody has read base / s_off before they change
        if (threadIdx.x == 0) base = b0 + s_off[16];
    }
    __syncthreads();
    if (threadIdx.x == 0) flen[no] = base;
}
hipError_t launch_filter_lists(const int64_t *d_pl_off, const int32_t *d_pl_ids, int nlist,
                               const uint32_t *d_bitmap, int32_t *d_fids, int32_t *d_flen, hipStream_t st)
{
    if (nlist == 0) return hipSuccess;
    hipLaunchKernelGGL(filter_lists_kernel, dim3(nlist), dim3(256), 0, st, d_pl_off, d_pl_ids, d_bitmap, d_fids,
                       d_flen);
    return hipGetLastError();
}

// ===================================================================================================
// (a8) coarse assignment.  Symmetric tables D[m][k1][k2]; T_c[m][ks] = D[m][centre_c[m]][ks] (a row of D) so
// the assignment is the ADC scan with the centres in the role of queries and an arg-min over centres.
// ===================================================================================================
__global__ __launch_bounds__(256) void symtab_kernel(const float *__restrict__ cw, int M, int Ks, int Ds, int arch,
                                                     float *__restrict__ D)
{
    const int64_t total = (int64_t) M * Ks * Ks;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t) gridDim.x * blockDim.x) {
        const int k2 = (int) (t % Ks);
        const int64_t r = t / Ks;
        const int k1 = (int) (r % Ks);
        const int m = (int) (r / Ks);
        D[t] = l2sq_pqk_dev(cw + ((size_t) m * Ks + k1) * Ds, cw + ((size_t) m * Ks + k2) * Ds, Ds, arch);
    }
}
hipError_t launch_symtab(const float *d_codewords, int M, int Ks, int Ds, int arch, float *d_symtab, hipStream_t st)
{
    const int64_t total = (int64_t) M * Ks * Ks;
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(symtab_kernel, dim3(blocks), dim3(256), 0, st, d_codewords, M, Ks, Ds, arch, d_symtab);
    return hipGetLastError();
}

constexpr int kAssignThreads = 512;
constexpr int kAssignCPT = 4;     // codes per thread

template <int QT>
__global__ __launch_bounds__(kAssignThreads) void assign_kernel(const uint8_t *__restrict__ codes, int64_t num,
                                                                int M, int Ks, const float *__restrict__ D,
                                                                const uint8_t *__restrict__ centers, int nlist,
                                                                int32_t *__restrict__ assign)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename LutT<QT>::T LT;
    float *lds = reinterpret_cast<float *>(smem);
    const LT *lut = reinterpret_cast<const LT *>(lds);
    const int tid = threadIdx.x;
    const int MK = M * Ks;
    const int64_t base = (int64_t) blockIdx.x * (kAssignThreads * kAssignCPT);
    float bestd[kAssignCPT];
    int32_t besti[kAssignCPT];
#pragma unroll
    for (int k = 0; k < kAssignCPT; ++k) { bestd[k] = FLT_MAX; besti[k] = -1; }

    for (int c0 = 0; c0 < nlist; c0 += QT) {
        __syncthreads();
        // stage T for centres c0..c0+QT-1: lds[(m*Ks+ks)*QT + q] = D[m][centre[c0+q][m]][ks]
        for (int i = tid; i < MK * QT; i += kAssignThreads) {
            const int q = i % QT;
            const int mk = i / QT;
            const int m = mk / Ks, ks = mk - m * Ks;
            const int c = c0 + q;
            float v = 0.f;
            if (c < nlist) v = D[((size_t) m * Ks + centers[(size_t) c * M + m]) * Ks + ks];
            lds[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kAssignCPT; ++k) {
            const int64_t n = base + (int64_t) k * kAssignThreads + tid;
            if (n >= num) continue;
            const uint8_t *code = codes + (size_t) n * M;
            float acc[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[q] = 0.f;
            if ((M & 3) == 0) {
                const uint32_t *cw = reinterpret_cast<const uint32_t *>(code);
                for (int i = 0; i < M / 4; ++i) {
                    const uint32_t w = cw[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc_add<QT>(acc, lut[(i * 4 + j) * Ks + ((w >> (8 * j)) & 0xffu)]);
                }
            } else {
                for (int m = 0; m < M; ++m) acc_add<QT>(acc, lut[m * Ks + code[m]]);
            }
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                // ascending centre index + strict '<' == first minimum (src/pqkmeans.cpp:209-215)
                if (c0 + q < nlist && acc[q] < bestd[k]) { bestd[k] = acc[q]; besti[k] = c0 + q; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kAssignCPT; ++k) {
        const int64_t n = base + (int64_t) k * kAssignThreads + tid;
        if (n < num) assign[n] = besti[k];
    }
}

template <int QT>
static hipError_t launch_assign_t(const uint8_t *d_codes, int64_t num, int M, int Ks, const float *d_symtab,
                                  const uint8_t *d_centers, int nlist, int32_t *d_assign, hipStream_t st)
{
    const size_t smem = (size_t) M * Ks * QT * sizeof(float);
    auto kern = assign_kernel<QT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    const int64_t per_block = (int64_t) kAssignThreads * kAssignCPT;
    hipLaunchKernelGGL(kern, dim3((unsigned) ((num + per_block - 1) / per_block)), dim3(kAssignThreads), smem, st,
                       d_codes, num, M, Ks, d_symtab, d_centers, nlist, d_assign);
    return hipGetLastError();
}
hipError_t launch_assign(const uint8_t *d_codes, int64_t num, int M, int Ks, const float *d_symtab,
                         const uint8_t *d_centers, int nlist, int32_t *d_assign, hipStream_t st)
{
    if (num == 0) return hipSuccess;
    const int QT = lut_tile_for(M, Ks);
    if (QT == 0) return launch_assign_wide(d_codes, num, M, Ks, d_symtab, d_centers, nlist, d_assign, st);
    if (QT == 4) return launch_assign_t<4>(d_codes, num, M, Ks, d_symtab, d_centers, nlist, d_assign, st);
    if (QT == 2) return launch_assign_t<2>(d_codes, num, M, Ks, d_symtab, d_centers, nlist, d_assign, st);
    return launch_assign_t<1>(d_codes, num, M, Ks, d_symtab, d_centers, nlist, d_assign, st);
}

// ===================================================================================================
// (f1) PQk-means centre update (src/pqkmeans.cpp:109-123, 223-260): per (cluster, m) histogram of the
// assigned codes' bytes, distance-weighted vote with one FMA per (k1,k2) in ascending k1, first arg-min.
// ===================================================================================================
__global__ void pqk_hist_kernel(const uint8_t *__restrict__ data, const int32_t *__restrict__ assign, int64_t n,
                                int M, int Ks, int32_t *__restrict__ hist, int32_t *__restrict__ cnt)
{
    const int64_t total = n * M;
    for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i = t / M;
        const int m = (int) (t - i * M);
        const int k = assign[i];
        atomicAdd(&hist[((size_t) k * M + m) * Ks + data[t]], 1);
        if (m == 0) atomicAdd(&cnt[k], 1);
    }
}
hipError_t launch_pqk_hist(const uint8_t *d_data, const int32_t *d_assign, int64_t n, int M, int Ks,
                           int32_t *d_hist, int32_t *d_cnt, hipStream_t st)
{
    const int64_t total = n * M;
    if (total == 0) return hipSuccess;
    int blocks = (int) ((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pqk_hist_kernel, dim3(blocks), dim3(256), 0, st, d_data, d_assign, n, M, Ks, d_hist, d_cnt);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pqk_vote_kernel(const int32_t *__restrict__ hist,
                                                       const int32_t *__restrict__ cnt,
                                                       const float *__restrict__ D, int K, int M, int Ks,
                                                       uint8_t *__restrict__ centers)
{
    __shared__ int sh_hist[256];
    __shared__ unsigned long long red;
    const int k = blockIdx.x / M, m = blockIdx.x - k * M;
    if (cnt[k] == 0) return;                                   // empty cluster keeps its old centre
    const int k2 = threadIdx.x;
    if (k2 < Ks) sh_hist[k2] = hist[((size_t) k * M + m) * Ks + k2];
    if (threadIdx.x == 0) red = ~0ull;
    __syncthreads();
    unsigned long long key = ~0ull;
    if (k2 < Ks) {
        float vote = 0.f;
        for (int k1 = 0; k1 < Ks; ++k1) {
            const int f = sh_hist[k1];
            if (f == 0) continue;
            vote = __fmaf_rn((float) f, D[((size_t) m * Ks + k1) * Ks + k2], vote);
        }
        // first minimum under strict '<' against FLT_MAX (pqkmeans.cpp:248-256): votes >= FLT_MAX never win
        if (vote < FLT_MAX) key = ((unsigned long long) f32_orderable(__float_as_uint(vote)) << 32) | (uint32_t) k2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off);
        key = o < key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key != ~0ull) atomicMin(&red, key);
    __syncthreads();
    if (threadIdx.x == 0 && red != ~0ull) centers[(size_t) k * M + m] = (uint8_t) (red & 0xffu);
}
hipError_t launch_pqk_vote(const int32_t *d_hist, const int32_t *d_cnt, const float *d_symtab, int K, int M,
                           int Ks, uint8_t *d_centers, hipStream_t st)
{
    if (K == 0) return hipSuccess;
    hipLaunchKernelGGL(pqk_vote_kernel, dim3((unsigned) (K * M)), dim3(256), 0, st, d_hist, d_cnt, d_symtab, K, M,
                       Ks, d_centers);
    return hipGetLastError();
}

}  // namespace riiamd
