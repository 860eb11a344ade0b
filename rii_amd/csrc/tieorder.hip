// tieorder.hip -- the reference's result order for linear top-k (k > 1) when distances tie exactly (gfx950).
//
// RiiCpp::QueryLinear (src/rii.h:234-235) selects with std::partial_sort on the distance alone.  Whenever the k+1
// smallest distances are pairwise different its outcome is the canonical (dist, id) order the filter + re-rank
// kernels of fastscan.hip produce.  When two of them are bit-equal, which of the tied codes is returned and in which
// order is decided by libstdc++'s heap (__make_heap over the first k scores, __heap_select over the rest in index
// order, __sort_heap).  The producers of the canonical result raise a per-query flag in exactly that case and this
// kernel redoes the flagged queries by replaying the library's algorithm over the only elements that can touch the
// heap:
//
//   element i >= k changes the heap iff d_i < (heap top when i is visited), and the heap top never rises.  So for any
//   upper bound thr >= top the elements with d_i < thr are a superset of the ones that enter, and replaying
//   __heap_select over a superset *in index order* is move-for-move the full run (the extra elements fail the very
//   comparison the library makes, `comp(*i, *first)`, and leave the heap untouched).
//
// One block per flagged query: all 256 threads evaluate exact fp32 distances (RiiCpp::ADist order) of a slab of codes
// and keep (index, dist) pairs below the last known heap top in an LDS list; whenever the list could overflow it is
// sorted by index and ONE lane feeds it to the heap, which refreshes the bound.  About k (1 + ln(n/k)) elements ever
// pass, so the serial part stays short; the scan itself is the exhaustive ADC of one query.
#include "rii_internal.h"
#include "rii_device.h"
#include <algorithm>

namespace riiamd {

// LDS list of pending (index, dist) pairs: 2048 entries (a trip of four codes per thread appends at most 1024), or 512
// (one code per thread and trip) next to the largest tables
static int tie_list_cap(int M, int Ks) { return (size_t) M * Ks * 4 + 2049 * 8 + 64 <= (size_t) 160 * 1024 - 512 ? 2048 : 512; }

struct TieArgs {
    const uint8_t *codes;                // [n][M] in INDEX order of the reference's `scores` array (ids, or tids positions)
    int64_t n;
    int M, Ks;
    const float *lut; int QT;            // exact tables of the batch, lut_index layout
    int64_t b0;                          // flagged entries are query indices relative to b0 (tables / outputs at b0 + f)
    const int32_t *flag_list; const int *nflag;
    const int64_t *remap;                // index -> id (subset search), or NULL
    int64_t *out_ids; float *out_dists; int topk;
    int32_t *g_hid; float *g_hd;         // [gridDim.x][topk] heap storage when it does not fit LDS (else unused)
    int heap_in_lds;
    int capl;                            // list capacity (tie_list_cap)
    int indirect;                        // 1: `codes` is the whole database, index i stands for the code remap[i]
};

__device__ __forceinline__ void tie_flush(unsigned long long *list, unsigned int c, int32_t *hid, float *hd, long k,
                                          float *s_thr, unsigned int *s_cnt, int tid)
{
    // uniform: c was read between two barriers
    if (c) {
        int nsort = 64;
        while (nsort < (int) c) nsort <<= 1;
        for (int i = tid; i < nsort; i += 256)
            if ((unsigned int) i >= c) list[i] = ~0ull;
        rr_bitonic_sort(list, tid, nsort);                 // ascending (index << 32 | dist bits): index order
        if (tid == 0) {
            for (unsigned int j = 0; j < c; ++j) {         // __heap_select, bits/stl_algo.h
                const unsigned long long e = list[j];
                const float d = __uint_as_float((uint32_t) (e & 0xffffffffu));
                if (d < hd[0]) pq_adjust_heap(hid, hd, 0, k, (int32_t) (e >> 32), d);     // __pop_heap(first, middle, i)
            }
            *s_thr = hd[0];
            *s_cnt = 0u;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void linear_tie_kernel(TieArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int MK = p.M * p.Ks;
    float *lds = reinterpret_cast<float *>(smem);
    unsigned long long *list = reinterpret_cast<unsigned long long *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));
    float *s_thr = reinterpret_cast<float *>(list + p.capl);
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_thr + 1);
    const long k = p.topk;
    int32_t *hid;
    float *hd;
    if (p.heap_in_lds) {
        hd = reinterpret_cast<float *>(list + p.capl + 1);
        hid = reinterpret_cast<int32_t *>(hd + k);
    } else {
        hd = p.g_hd + (size_t) blockIdx.x * k;
        hid = p.g_hid + (size_t) blockIdx.x * k;
    }
    const int tid = threadIdx.x;
    const int nflag = *p.nflag;
    for (int fi = blockIdx.x; fi < nflag; fi += gridDim.x) {
        __syncthreads();                                   // the previous query's LDS contents are dead from here on
        const int64_t b = p.b0 + p.flag_list[fi];
        {
            const float *src = p.lut + (size_t) (b / p.QT) * MK * p.QT + (b % p.QT);
            for (int i = tid; i < MK; i += 256) lds[i] = src[(size_t) i * p.QT];
        }
        __syncthreads();
        for (long i = tid; i < k; i += 256) {              // the first k scores are the initial heap contents
            hd[i] = exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[i] : (int64_t) i) * p.M, p.M, p.Ks);
            hid[i] = (int32_t) i;
        }
        if (!p.heap_in_lds) __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            if (k >= 2) {                                  // __make_heap, bits/stl_heap.h
                long parent = (k - 2) / 2;
                for (;;) {
                    pq_adjust_heap(hid, hd, parent, k, hid[parent], hd[parent]);
                    if (parent == 0) break;
                    parent--;
                }
            }
            *s_thr = hd[0];
            *s_cnt = 0u;
        }
        const int U = p.capl / 512;                        // codes per thread and trip: 4 or 1
        for (int64_t base = k; base < p.n; base += (int64_t) U * 256) {
            __syncthreads();                               // appends of the previous trip are complete ...
            const unsigned int c = *s_cnt;
            __syncthreads();                               // ... and everybody saw the same count before the next ones
            if (c + (unsigned int) U * 256u > (unsigned int) p.capl) tie_flush(list, c, hid, hd, k, s_thr, s_cnt, tid);
            const float thr = *s_thr;
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * 256 + tid;
                d[u] = (u < U && i < p.n) ? exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[i] : (int64_t) i) * p.M, p.M, p.Ks) : INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * 256 + tid;
                if (u < U && i < p.n && d[u] < thr)
                    list[atomicAdd(s_cnt, 1u)] = ((unsigned long long) (uint32_t) i << 32) | __float_as_uint(d[u]);
            }
        }
        __syncthreads();
        const unsigned int c = *s_cnt;
        __syncthreads();
        tie_flush(list, c, hid, hd, k, s_thr, s_cnt, tid);
        if (tid == 0) {                                    // __sort_heap
            long len = k;
            while (len > 1) {
                --len;
                const int32_t vid = hid[len];
                const float vd = hd[len];
                hid[len] = hid[0]; hd[len] = hd[0];
                pq_adjust_heap(hid, hd, 0, len, vid, vd);
            }
            if (!p.heap_in_lds) __threadfence_block();
        }
        __syncthreads();
        for (long j = tid; j < k; j += 256) {
            const int32_t idx = hid[j];
            p.out_ids[b * k + j] = p.remap ? p.remap[idx] : (int64_t) idx;
            p.out_dists[b * k + j] = hd[j];
        }
    }
}

static size_t tie_smem(int M, int Ks, int topk, bool heap_in_lds)
{
    return (((size_t) M * Ks * 4 + 15) & ~(size_t) 15) + (size_t) (tie_list_cap(M, Ks) + 1) * 8 + (heap_in_lds ? (size_t) topk * 8 : 0) + 16;
}
bool linear_tie_heap_in_lds(int M, int Ks, int topk) { return tie_smem(M, Ks, topk, true) <= (size_t) 160 * 1024 - 512; }
bool linear_tie_supported(int M, int Ks) { return tie_smem(M, Ks, 0, false) <= (size_t) 160 * 1024 - 512; }

hipError_t launch_linear_tie(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, int QT, int64_t b0,
                             const int32_t *d_flag_list, const int *d_nflag, const int64_t *d_remap, int64_t *d_out_ids,
                             float *d_out_dists, int topk, int grid, int32_t *d_heap_ids, float *d_heap_d, int indirect, hipStream_t st)
{
    TieArgs a;
    a.codes = d_codes; a.n = n; a.M = M; a.Ks = Ks; a.lut = d_lut; a.QT = QT; a.b0 = b0; a.flag_list = d_flag_list;
    a.nflag = d_nflag; a.remap = d_remap; a.out_ids = d_out_ids; a.out_dists = d_out_dists; a.topk = topk;
    a.g_hid = d_heap_ids; a.g_hd = d_heap_d;
    a.heap_in_lds = linear_tie_heap_in_lds(M, Ks, topk) ? 1 : 0;
    a.capl = tie_list_cap(M, Ks);
    a.indirect = indirect;
    const size_t smem = tie_smem(M, Ks, topk, a.heap_in_lds != 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(linear_tie_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(linear_tie_kernel, dim3((unsigned) grid), dim3(256), smem, st, a);
    return hipGetLastError();
}

}  // namespace riiamd
