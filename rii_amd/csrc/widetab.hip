// widetab.hip -- shapes whose ONE-query distance table (M * Ks * 4 bytes) does not fit the LDS budget (gfx950, round 3).
//
// The reference takes any M <= D and Ks <= 256 (rii/rii.py:35, src/rii.h:361-373): M = 160 at Ks = 256 is a 160 KiB table, M = 256 a
// 256 KiB one; every kernel of the main path stages at least one whole table in LDS (lut_tile_for() == 0 for these shapes).
// Here the exact fp32 tables stay in global memory (L2-resident: B * M * Ks * 4 bytes per batch) and
//   * scan_wide_kernel    walks the table in SLICES of subspaces that fit LDS, every thread carrying the running sums of its
//                         codes across the slices -- the additions still happen in the reference's m order (RiiCpp::ADist,
//                         src/rii.h:386-394) -- and emits one packed (distance, index) key per (query, code);
//   * the keys go through the general top-k machinery of the engine (segmented sort -> (dist, id) order, exact ties flagged);
//   * tie_rows_kernel     replays std::partial_sort (src/rii.h:234-235) over the key row of a flagged query: the row IS the
//                         reference's `scores` array in index order (heap in LDS, the rest streamed: wh_partial_sort_split);
//   * assign_wide_kernel  coarse assignment (PQKMeans::predict_one, src/pqkmeans.cpp:193-218) straight from the symmetric tables in
//                         global memory.
// The inverted index of these shapes runs ivf_exact_big_kernel<GTAB> (kernels.hip) for every query.  Correctness first: these
// kernels are not tuned; the filter scan is not used.
#include "rii_internal.h"
#include "rii_device.h"
#include <float.h>
#include <algorithm>

namespace riiamd {

constexpr int kWideThreads = 256;
constexpr int kWideCPT = 8;                       // codes per thread: one block covers 2048 codes
constexpr size_t kWideSliceBytes = 128 * 1024;    // table slice staged in LDS

struct WideArgs {
    const uint8_t *codes; int64_t n_codes; int M, Ks;
    const float *lut;                 // plain [b][M * Ks]
    const int64_t *remap;             // subset search: position n stands for the code remap[n] (or NULL)
    int b0;                           // first query of this launch inside the batch
    int ms;                           // subspaces per slice
    unsigned long long *keys;         // [gridDim.y][n_codes]
};

__global__ __launch_bounds__(kWideThreads) void scan_wide_kernel(WideArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lds = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x;
    const int64_t bq = blockIdx.y;                                     // query inside the launch
    const int64_t base = (int64_t) blockIdx.x * (kWideThreads * kWideCPT);
    const float *tab = p.lut + (size_t) (p.b0 + bq) * p.M * p.Ks;
    float acc[kWideCPT];
    const uint8_t *row[kWideCPT];
#pragma unroll
    for (int j = 0; j < kWideCPT; ++j) {
        acc[j] = 0.f;
        const int64_t n = base + (int64_t) j * kWideThreads + tid;
        row[j] = n < p.n_codes ? p.codes + (size_t) (p.remap ? p.remap[n] : n) * p.M : nullptr;
    }
    for (int m0 = 0; m0 < p.M; m0 += p.ms) {
        const int mc = (p.M - m0 < p.ms) ? p.M - m0 : p.ms;
        __syncthreads();                                               // the previous slice is dead
        for (int i = tid; i < mc * p.Ks; i += kWideThreads) lds[i] = tab[(size_t) m0 * p.Ks + i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kWideCPT; ++j) {
            if (!row[j]) continue;
            float a = acc[j];
            for (int m = 0; m < mc; ++m) a = __fadd_rn(a, lds[m * p.Ks + row[j][m0 + m]]);     // sequential over m, across slices
            acc[j] = a;
        }
    }
#pragma unroll
    for (int j = 0; j < kWideCPT; ++j) {
        const int64_t n = base + (int64_t) j * kWideThreads + tid;
        if (n < p.n_codes)
            p.keys[(size_t) bq * p.n_codes + n] = ((unsigned long long) f32_orderable(__float_as_uint(acc[j])) << 32) | (uint32_t) n;
    }
}

hipError_t launch_scan_wide(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const float *d_lut, const int64_t *d_remap,
                            int b0, int bc, unsigned long long *d_keys, hipStream_t st)
{
    if (bc == 0 || n_codes == 0) return hipSuccess;
    WideArgs a;
    a.codes = d_codes; a.n_codes = n_codes; a.M = M; a.Ks = Ks; a.lut = d_lut; a.remap = d_remap; a.b0 = b0; a.keys = d_keys;
    a.ms = (int) std::max<size_t>(1, std::min<size_t>((size_t) M, kWideSliceBytes / ((size_t) Ks * 4)));
    const size_t smem = (size_t) a.ms * Ks * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(scan_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    const int64_t per = (int64_t) kWideThreads * kWideCPT;
    hipLaunchKernelGGL(scan_wide_kernel, dim3((unsigned) ((n_codes + per - 1) / per), (unsigned) bc), dim3(kWideThreads), smem, st, a);
    return hipGetLastError();
}

// ---- exact ties: std::partial_sort over the key row of a flagged query (a small persistent grid walks the flag list) ----
__global__ __launch_bounds__(64) void tie_rows_kernel(unsigned long long *__restrict__ keys, int64_t n_codes, int64_t b0,
                                                      const int32_t *__restrict__ flag_list, const int *__restrict__ nflag, int topk,
                                                      const int64_t *__restrict__ remap, int64_t *__restrict__ out_ids,
                                                      float *__restrict__ out_dists)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pq64_t *s_head = reinterpret_cast<pq64_t *>(smem);                 // [topk] when the heap fits the wave code
    const int lane = threadIdx.x;
    const int nf = *nflag;
    const bool k_lds = topk <= kWhSplitMaxHeap;
    for (int fi = blockIdx.x; fi < nf; fi += gridDim.x) {
        const int64_t q = flag_list[fi];                               // row inside this launch group
        pq64_t *row = keys + (size_t) q * n_codes;                     // (orderable distance << 32 | index), index order
        const int k = topk < n_codes ? topk : (int) n_codes;
        if (k_lds) {
            for (int j = lane; j < k; j += 64) s_head[j] = row[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            wh_partial_sort_split(s_head, row + k, k, (int) n_codes, lane);
            __builtin_amdgcn_wave_barrier();
        } else {
            if (lane == 0) pq64_partial_sort(row, k, (long) n_codes);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        for (int j = lane; j < k; j += 64) {
            const pq64_t e = k_lds ? s_head[j] : row[j];
            const uint32_t idx = pq64_id(e);
            out_ids[(b0 + q) * topk + j] = remap ? remap[idx] : (int64_t) idx;
            out_dists[(b0 + q) * topk + j] = pq64_dist(e);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launch_tie_rows(unsigned long long *d_keys, int64_t n_codes, int64_t b0, int64_t bc, const int32_t *d_flag_list, const int *d_nflag,
                           int topk, const int64_t *d_remap, int64_t *d_out_ids, float *d_out_dists, hipStream_t st)
{
    if (bc == 0 || topk < 2) return hipSuccess;
    const size_t smem = topk <= kWhSplitMaxHeap ? (size_t) topk * 8 : 16;
    hipLaunchKernelGGL(tie_rows_kernel, dim3((unsigned) std::min<int64_t>(bc, 256)), dim3(64), smem, st, d_keys, n_codes, b0, d_flag_list,
                       d_nflag, topk, d_remap, d_out_ids, d_out_dists);
    return hipGetLastError();
}

// ---- coarse assignment from the symmetric tables in global memory: argmin_c sum_m D[m][centre_c[m]][code_m], first minimum ----
__global__ __launch_bounds__(256) void assign_wide_kernel(const uint8_t *__restrict__ codes, int64_t num, int M, int Ks,
                                                          const float *__restrict__ D, const uint8_t *__restrict__ centers, int nlist,
                                                          int32_t *__restrict__ assign)
{
    const int64_t n = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= num) return;
    const uint8_t *code = codes + (size_t) n * M;
    float bestd = FLT_MAX;
    int32_t besti = -1;
    for (int c = 0; c < nlist; ++c) {
        const uint8_t *cen = centers + (size_t) c * M;
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc = __fadd_rn(acc, D[((size_t) m * Ks + cen[m]) * Ks + code[m]]);
        if (acc < bestd) { bestd = acc; besti = c; }                    // ascending c + strict '<' == first minimum (pqkmeans.cpp:209-215)
    }
    assign[n] = besti;
}

hipError_t launch_assign_wide(const uint8_t *d_codes, int64_t num, int M, int Ks, const float *d_symtab, const uint8_t *d_centers, int nlist,
                              int32_t *d_assign, hipStream_t st)
{
    if (num == 0) return hipSuccess;
    hipLaunchKernelGGL(assign_wide_kernel, dim3((unsigned) ((num + 255) / 256)), dim3(256), 0, st, d_codes, num, M, Ks, d_symtab, d_centers,
                       nlist, d_assign);
    return hipGetLastError();
}

}  // namespace riiamd
