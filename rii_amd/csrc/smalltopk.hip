// smalltopk.hip -- QueryLinear (src/rii.h:195-242) for a SMALL index and a SMALL batch, in ONE launch, tables included (gfx950, round 3).
//
// The reference's own usage pattern is one query per call on an index of ~10^4 codes (README.md:84-140: N = 10 000, topk = 3).  The
// batched top-k path spends eight launches there (tables, two filter passes, k-th threshold, re-rank, three tie-order kernels): 90 us
// per call against 97 us for the reference on one CPU core.  Here one block per query holds the query's exact table AND the exact
// distance of every code, in index order, in LDS -- the reference's `scores` array -- and
//   1. bounds the (k+1)-th smallest distance from above with a 256-bin histogram of the occupied range (refined bin by bin only
//      while more than 2048 keys lie under the bound), sorts the few keys up to it under (dist, id);
//   2. if no two of the k+1 smallest distances are bit-equal that IS std::partial_sort's answer;
//   3. else one wave replays std::partial_sort (rii_device.h: wh_partial_sort) over the array itself -- the library's algorithm on the
//      library's input, move for move.
// n <= (160 KiB - table - 19.5 KiB) / 8 codes (13 800 at M = 32, Ks = 256), batches below the filter's crossover (fast_min_batch).
#include "rii_internal.h"
#include "rii_device.h"
#include <algorithm>

namespace riiamd {

constexpr int kStThreads = 1024;
constexpr int kStBuf = 2048;                 // keys up to the bound that are sorted (more: exact ties galore -> replay)
constexpr int kStRank = 256;                 // up to this many keys are sorted by counting

struct SmallArgs {
    const uint8_t *codes; int64_t n; int M, Ks;
    const float *lut;                        // plain [b][M * Ks], or NULL: the block builds its table itself from
    const float *queries, *codewords;        //   queries [b][M * Ds] and the codebook (RiiCpp::DTable, src/rii.h:361-373)
    int Ds, arch;                            //   with the fvec_L2sqr flavour `arch` (src/distance.h:117-252)
    const int64_t *remap;                    // subset search: index i stands for the code remap[i] (ids translated on output), or NULL
    int topk;
    int64_t *out_ids; float *out_dists;
    unsigned long long *gkeys;               // gridDim.x > 1: [b][n] keys of all slices (the last block to finish selects)
    unsigned int *done;                      // [b] blocks of the query that have stored their slice; back to 0 when the launch ends
    unsigned int *host_flag; unsigned int seq;   // out_ids / out_dists are coherent HOST memory: flag[b] = seq once query b's rows are
                                                 // written (system-scope release); NULL: ordinary device outputs
};

// ascending bitonic sort of n (a power of two) 64-bit keys in LDS by all threads of the block
__device__ __forceinline__ void st_bitonic(unsigned long long *buf, int tid, int n)
{
    for (int size = 2; size <= n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < n / 2; t += kStThreads) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = ((i & size) == 0);
                const unsigned long long x = buf[i], y = buf[j];
                if ((x > y) == up) { buf[i] = y; buf[j] = x; }
            }
        }
    __syncthreads();
}

// one code row of NV * 16 bytes in registers; the sum in m order (RiiCpp::ADist, src/rii.h:386-394)
template <int NV>
__device__ __forceinline__ float st_adist(const float *lds, const uint4 (&w)[NV], int Ks)
{
    float d = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const uint32_t x[4] = {w[v].x, w[v].y, w[v].z, w[v].w};
#pragma unroll
        for (int j = 0; j < 16; ++j) d = __fadd_rn(d, lds[(v * 16 + j) * Ks + ((x[j >> 2] >> (8 * (j & 3))) & 0xffu)]);
    }
    return d;
}

// The exact table of ONE query (RiiCpp::DTable, src/rii.h:361-373) built by the block into LDS (no barrier after the last store: the
// caller's next barrier covers it).  The query goes through LDS: it is read from memory ONCE per block -- it may be the caller's
// pinned HOST block (host_spin: no H2D copy in front of the launch; every read of it crosses PCIe).  Its load is requested first,
// parked in a register while the first codebook entries are requested too, and only then stored and waited for.
__device__ __forceinline__ void st_build_table(const float *qg, const float *codewords, int M, int Ks, int Ds, int arch, float *lds, float *s_q)
{
    const int tid = threadIdx.x, MK = M * Ks, D = M * Ds;
    float qreg[4] = {0.f, 0.f, 0.f, 0.f};
    if ((D & 3) == 0) {
        if (tid < D / 4) {
            const float4 v = reinterpret_cast<const float4 *>(qg)[tid];       // D <= 4096 floats: one 16-byte piece per thread
            qreg[0] = v.x; qreg[1] = v.y; qreg[2] = v.z; qreg[3] = v.w;
        }
    }
    if (Ds == 4) {
        // Ds == 4: straight-line fvec_L2sqr, identical for the three SIMD flavours
        const float4 *cw4 = reinterpret_cast<const float4 *>(codewords);
        const float4 *q4 = reinterpret_cast<const float4 *>(s_q);
        const int sh = (Ks & (Ks - 1)) == 0 ? __ffs(Ks) - 1 : -1;
        for (int i0 = 0; i0 < MK; i0 += 8 * kStThreads) {
            float4 cv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = i0 + u * kStThreads + tid, ic = idx < MK ? idx : MK - 1;
                cv[u] = cw4[ic];
            }
            if (i0 == 0) {
                if (tid < D / 4) reinterpret_cast<float4 *>(s_q)[tid] = make_float4(qreg[0], qreg[1], qreg[2], qreg[3]);
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = i0 + u * kStThreads + tid;
                if (idx < MK) lds[idx] = fvec_l2sqr_ds4v(q4[sh >= 0 ? idx >> sh : idx / Ks], cv[u]);
            }
        }
    } else {
        if ((D & 3) == 0) {
            if (tid < D / 4) reinterpret_cast<float4 *>(s_q)[tid] = make_float4(qreg[0], qreg[1], qreg[2], qreg[3]);
        } else {
            for (int i = tid; i < D; i += kStThreads) s_q[i] = qg[i];
        }
        __syncthreads();
        for (int i = tid; i < MK; i += kStThreads) lds[i] = fvec_l2sqr_any(s_q + (i / Ks) * Ds, codewords + (size_t) i * Ds, Ds, arch);
    }
}

// NV > 0: M == 16 * NV, code rows loaded as NV 16-byte words.  NV == 0: any M.
// One block of 1024 threads per (slice, query).  Everything here is a chain of dependent memory round trips (~1 us each at one block
// per CU), so every phase issues all its loads before it uses any: the code rows of a thread's first codes are requested BEFORE the
// table is built, the codebook entries of a thread eight at a time, the keys of the other slices eight at a time.
template <int NV>
__global__ __launch_bounds__(kStThreads) void small_topk_kernel(SmallArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int MK = p.M * p.Ks, tid = threadIdx.x, lane = tid & 63, n = (int) p.n, k = p.topk;
    float *lds = reinterpret_cast<float *>(smem);
    float *s_q = reinterpret_cast<float *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));          // [M * Ds] the query (16-byte aligned)
    pq64_t *s_key = reinterpret_cast<pq64_t *>(reinterpret_cast<unsigned char *>(s_q) + (((size_t) p.M * p.Ds * 4 + 15) & ~(size_t) 15));   // [n] (distance, index), index order
    unsigned long long *s_buf = s_key + n;                                                           // [kStBuf] keys up to the bound
    unsigned long long *s_out = s_buf + kStBuf;                                                      // [kStRank] rank-sorted
    unsigned int *s_hist = reinterpret_cast<unsigned int *>(s_out + kStRank);                        // [256]
    unsigned int *s_ctl = s_hist + 256;                                                              // [8]
    const int64_t b = blockIdx.y;
    const int G = (int) gridDim.x;
    const int per = (n + G - 1) / G;                                   // slice of this block
    const int s_lo = (int) blockIdx.x * per < n ? (int) blockIdx.x * per : n, s_hi = s_lo + per < n ? s_lo + per : n;
    pq64_t *dst = G > 1 ? p.gkeys + (size_t) b * n : s_key;
    constexpr int PRE = NV == 0 ? 0 : NV <= 2 ? 4 : 2;                 // codes per thread requested before the table exists
    uint4 pre[PRE * NV + 1];
    if constexpr (NV > 0) {
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int i = s_lo + tid + c * kStThreads, ic = i < s_hi ? i : n - 1;
            const int64_t r = p.remap ? p.remap[ic] : (int64_t) ic;
            const uint4 *row = reinterpret_cast<const uint4 *>(p.codes + (size_t) r * (NV * 16));
#pragma unroll
            for (int v = 0; v < NV; ++v) pre[c * NV + v] = row[v];
        }
    }
    if (tid == 0) s_ctl[5] = 0u;
    if (!p.lut) {
        st_build_table(p.queries + (size_t) b * p.M * p.Ds, p.codewords, p.M, p.Ks, p.Ds, p.arch, lds, s_q);
    } else {
        const float *src = p.lut + (size_t) b * MK;
        if ((MK & 3) == 0) {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(lds);
            for (int i = tid; i < MK / 4; i += kStThreads) d4[i] = s4[i];
        } else {
            for (int i = tid; i < MK; i += kStThreads) lds[i] = src[i];
        }
    }
    __syncthreads();
    if constexpr (NV > 0) {
#pragma unroll
        for (int c = 0; c < PRE; ++c) {
            const int i = s_lo + tid + c * kStThreads;
            uint4 w[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) w[v] = pre[c * NV + v];
            if (i < s_hi) {
                const pq64_t e = pq64_make(st_adist<NV>(lds, w, p.Ks), (uint32_t) i);                 // RiiCpp::ADist, m order
                dst[i] = e;
            }
        }
        for (int i = s_lo + tid + PRE * kStThreads; i < s_hi; i += 2 * kStThreads) {                   // the rest two at a time
            const int i2 = i + kStThreads < s_hi ? i + kStThreads : i;
            const int64_t r1 = p.remap ? p.remap[i] : (int64_t) i, r2 = p.remap ? p.remap[i2] : (int64_t) i2;
            const uint4 *row1 = reinterpret_cast<const uint4 *>(p.codes + (size_t) r1 * (NV * 16));
            const uint4 *row2 = reinterpret_cast<const uint4 *>(p.codes + (size_t) r2 * (NV * 16));
            uint4 w1[NV], w2[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) { w1[v] = row1[v]; w2[v] = row2[v]; }
            const pq64_t e1 = pq64_make(st_adist<NV>(lds, w1, p.Ks), (uint32_t) i);
            const pq64_t e2 = pq64_make(st_adist<NV>(lds, w2, p.Ks), (uint32_t) i2);
            dst[i] = e1;
            dst[i2] = e2;                              // i2 == i past the end: the same key again
        }
    } else {
        for (int i = s_lo + tid; i < s_hi; i += kStThreads) {
            const uint8_t *code = p.codes + (size_t) (p.remap ? p.remap[i] : (int64_t) i) * p.M;
            const pq64_t e = pq64_make(exact_adist(lds, code, p.M, p.Ks), (uint32_t) i);              // RiiCpp::ADist, m order
            dst[i] = e;
        }
    }
    if (G > 1) {
        // the slices of a query run on G CUs; the LAST block to store its slice gathers all keys into its LDS and selects.
        // Release / acquire at device scope around the counter (the slices sit in different XCDs' L2s).
        __syncthreads();                               // the block's keys are written (workgroup scope) ...
        if (tid == 0) {
            __threadfence();                           // ... and released to the device by ONE wave (an L2 write-back each)
            s_ctl[6] = atomicAdd(&p.done[b], 1u);
            __threadfence();
        }
        __syncthreads();
        if (s_ctl[6] != (unsigned int) (G - 1)) return;
        if (tid == 0) p.done[b] = 0u;                  // nobody else looks at it before the next launch
        const pq64_t *src = p.gkeys + (size_t) b * n;
        for (int i0 = 0; i0 < n; i0 += 8 * kStThreads) {
            pq64_t e[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kStThreads + tid;
                e[u] = __hip_atomic_load(src + (i < n ? i : n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past this XCD's L2
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * kStThreads + tid;
                if (i < n) s_key[i] = e[u];
            }
        }
    }
    // ---- a bound T with k+1 <= #{distance <= T} <= kStBuf: typically ONE histogram pass (the bin holding the (k+1)-th smallest of
    //      10^4 distances has a handful of keys below it) ----
    __syncthreads();
    const uint32_t k1 = (uint32_t) (k + 1 < n ? k + 1 : n);
    const uint32_t T = block_kth_bound([&](int i) { return (uint32_t) (s_key[i] >> 32); }, n, k1, (uint32_t) kStBuf, s_hist, s_ctl);
    for (int i0 = 0; i0 < n; i0 += kStThreads) {                    // keys up to the bound, appended wave by wave
        const int i = i0 + tid;
        const pq64_t e = i < n ? s_key[i] : ~0ull;
        const bool keep = i < n && (uint32_t) (e >> 32) <= T;
        const unsigned long long bal = __ballot(keep);
        if (bal) {
            unsigned int at0 = 0u;
            if (lane == 0) at0 = atomicAdd(&s_ctl[5], (unsigned int) __popcll(bal));
            at0 = (unsigned int) __shfl((int) at0, 0);
            const unsigned int at = at0 + (unsigned int) __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && at < (unsigned int) kStBuf) s_buf[at] = e;
        }
    }
    __syncthreads();
    const unsigned int nkeep = s_ctl[5];
    int tie = nkeep > (unsigned int) kStBuf ? 1 : 0;
    const unsigned long long *res = s_buf;
    if (!tie) {
        if (nkeep <= (unsigned int) kStRank) {
            // few keys: every key counts the keys under it ((dist, index) pairs are distinct) -- no barrier ladder
            if (tid < (int) nkeep) {
                const unsigned long long mine = s_buf[tid];
                unsigned int rank = 0u;
                for (unsigned int j = 0; j < nkeep; ++j) rank += s_buf[j] < mine ? 1u : 0u;
                s_out[rank] = mine;
            }
            res = s_out;
            __syncthreads();
        } else {
            int nsort = 512;
            while (nsort < (int) nkeep) nsort <<= 1;
            for (int i = tid; i < nsort; i += kStThreads)
                if ((unsigned int) i >= nkeep) s_buf[i] = ~0ull;
            st_bitonic(s_buf, tid, nsort);                           // (dist, index): the canonical order when nothing ties
        }
        for (int j = tid; j + 1 < (int) k1; j += kStThreads)
            if ((res[j] >> 32) == (res[j + 1] >> 32)) tie = 1;
    }
    if (__syncthreads_or(tie)) {
        // two of the k+1 smallest distances are bit-equal: the answer is whatever std::partial_sort (src/rii.h:234-235) makes of
        // the scores in index order -- replay it on the array itself
        if (tid < 64) wh_partial_sort(s_key, k, n, tid);
        __syncthreads();
        res = s_key;
    }
    for (int j = tid; j < k; j += kStThreads) {
        const pq64_t e = res[j];
        const uint32_t idx = pq64_id(e);
        p.out_ids[b * k + j] = p.remap ? p.remap[idx] : (int64_t) idx;
        p.out_dists[b * k + j] = pq64_dist(e);
    }
    if (p.host_flag) {
        // the rows went straight to the caller's pinned block: publish them to the host, which is spinning on the flag instead of
        // paying a D2H copy and a stream synchronisation (tools/host_latency_probe.hip: 9 us of a 17 us empty call)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every thread's row stores have left the CU before the barrier
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();
            __hip_atomic_store(&p.host_flag[b], p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static size_t small_topk_smem(int M, int Ks, int Ds, int64_t n)
{
    return (((size_t) M * Ks * 4 + 15) & ~(size_t) 15) + (size_t) n * 8 + (size_t) (kStBuf + kStRank) * 8 + 256 * 4 + 32 +
           (((size_t) M * Ds * 4 + 15) & ~(size_t) 15);
}
bool small_topk_supported(int M, int Ks, int Ds, int64_t n, int topk)
{
    return n >= 2 && topk >= 1 && topk <= n && topk + 1 <= kStBuf / 2 && (int64_t) M * Ds <= 4 * kStThreads &&
           small_topk_smem(M, Ks, Ds, n) <= (size_t) 160 * 1024 - 512;
}
int small_topk_slices(int64_t n, int64_t B)
{
    // one code per thread where the chip has the CUs for it (256 of them, one block each: 160 KiB of LDS)
    const int64_t by_n = (n + kStThreads - 1) / kStThreads, by_b = 256 / std::max<int64_t>(B, 1);
    return (int) std::max<int64_t>(1, std::min<int64_t>(16, std::min(by_n, by_b)));
}
size_t small_topk_scratch(int64_t n, int64_t B) { return small_topk_slices(n, B) > 1 ? (size_t) B * n * 8 : 0; }

hipError_t launch_small_topk(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, const float *d_queries,
                             const float *d_codewords, int Ds, int arch, int64_t B, int topk, const int64_t *d_remap,
                             unsigned long long *d_keys, unsigned int *d_done, int64_t *d_out_ids, float *d_out_dists, hipStream_t st,
                             unsigned int *host_flag, unsigned int seq)
{
    if (B == 0) return hipSuccess;
    SmallArgs a;
    a.codes = d_codes; a.n = n; a.M = M; a.Ks = Ks; a.lut = d_lut; a.queries = d_queries; a.codewords = d_codewords; a.Ds = Ds; a.arch = arch;
    a.remap = d_remap; a.topk = topk; a.out_ids = d_out_ids; a.out_dists = d_out_dists; a.gkeys = d_keys; a.done = d_done; a.host_flag = host_flag; a.seq = seq;
    const size_t smem = small_topk_smem(M, Ks, Ds, n);
    void (*kern)(SmallArgs) = small_topk_kernel<0>;
    if (M == 16) kern = small_topk_kernel<1>;
    else if (M == 32) kern = small_topk_kernel<2>;
    else if (M == 48) kern = small_topk_kernel<3>;
    else if (M == 64) kern = small_topk_kernel<4>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned) small_topk_slices(n, B), (unsigned) B), dim3(kStThreads), smem, st, a);
    return hipGetLastError();
}

// =====================================================================================================================
// ONE query per call on a LARGE index (round 3): the reference is used one query per call (rii/rii.py: Rii.query), and at N = 10^6
// the general top-k path spends eight launches on a single query (110 us for topk = 3; 43 us for topk = 1).  Host-pointer calls
// with a few queries (B <= 8) take ONE launch instead: grid (G slices, B queries), every block builds the query's exact table
// (st_build_table), scores its slice of the codes in passes of `pass_cap` codes -- keys (orderable distance << 32 | index) in LDS,
// the k+1 smallest under (distance, index) carried from pass to pass -- and writes its k+1 smallest keys to a global scratch; the
// LAST block to arrive at the per-query counter merges the G x (k+1) keys the same way and writes the rows.  If two of the k+1
// smallest distances of a query are bit-equal (the one case where std::partial_sort's answer is not the (distance, index) order),
// the kernel says so in out_tie[b] and the HOST reruns the call on the general path -- which is why only host-pointer calls (whose
// caller waits for the result anyway) come here.
// =====================================================================================================================
struct SliceArgs {
    const uint8_t *codes; int64_t n; int M, Ks;
    const float *queries, *codewords; int Ds, arch;
    const int64_t *remap;                    // subset search: index i stands for the code remap[i], or NULL
    int topk, k1max, pass_cap;               // k1max = topk + 1 (row pitch of cand)
    unsigned long long *cand;                // [B][G][k1max] the slices' smallest keys (padded with ~0)
    unsigned int *done;                      // [B] arrivals (low 16 bits) + "too many tied keys" marks (<< 16); zero between launches
    int64_t *out_ids; float *out_dists; int32_t *out_tie;
    unsigned int *host_flag; unsigned int seq;
    int32_t *flag_list = nullptr; int *nflag = nullptr;   // device-side tie fallback (asynchronous calls): tied queries appended here
};

// the kk smallest of keys[0, tot) in ascending (distance, index) order; *many: more than kStBuf keys lie under the bound (masses of
// bit-equal distances) -- the result is then only kk keys under the bound, not the smallest.  All threads call it.
__device__ __forceinline__ const unsigned long long *st_smallest(const pq64_t *keys, int tot, int kk, unsigned long long *s_buf,
                                                                 unsigned long long *s_out, unsigned int *s_hist, unsigned int *s_ctl, bool *many)
{
    const int tid = threadIdx.x, lane = tid & 63;
    __syncthreads();
    const uint32_t T = block_kth_bound([&](int i) { return (uint32_t) (keys[i] >> 32); }, tot, (uint32_t) kk, (uint32_t) kStBuf, s_hist, s_ctl);
    if (tid == 0) s_ctl[5] = 0u;
    __syncthreads();
    for (int i0 = 0; i0 < tot; i0 += kStThreads) {
        const int i = i0 + tid;
        const pq64_t e = i < tot ? keys[i] : ~0ull;
        const bool keep = i < tot && (uint32_t) (e >> 32) <= T;
        const unsigned long long bal = __ballot(keep);
        if (bal) {
            unsigned int at0 = 0u;
            if (lane == 0) at0 = atomicAdd(&s_ctl[5], (unsigned int) __popcll(bal));
            at0 = (unsigned int) __shfl((int) at0, 0);
            const unsigned int at = at0 + (unsigned int) __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && at < (unsigned int) kStBuf) s_buf[at] = e;
        }
    }
    __syncthreads();
    const unsigned int nkeep = s_ctl[5];
    *many = nkeep > (unsigned int) kStBuf;
    if (*many) return s_buf;
    if (nkeep <= (unsigned int) kStRank) {
        if (tid < (int) nkeep) {
            const unsigned long long mine = s_buf[tid];
            unsigned int rank = 0u;
            for (unsigned int j = 0; j < nkeep; ++j) rank += s_buf[j] < mine ? 1u : 0u;
            s_out[rank] = mine;
        }
        __syncthreads();
        return s_out;
    }
    int nsort = 512;
    while (nsort < (int) nkeep) nsort <<= 1;
    for (int i = tid; i < nsort; i += kStThreads)
        if ((unsigned int) i >= nkeep) s_buf[i] = ~0ull;
    st_bitonic(s_buf, tid, nsort);
    return s_buf;
}

// The kk (<= 16; used up to 6) smallest of the block's keys -- four per thread in registers, ~0 = none -- in ascending (distance, index) order into
// s_res[0, kk), without a histogram and with two barriers: every wave extracts its own kk smallest with wave-wide DPP minima
// (rii_device.h: wave_min_u64; keys are distinct, so "the smallest key greater than the last one" walks them in order), wave 0 then
// extracts the kk smallest of the 16 x kk picks the same way.  s_w: 16 x 16 keys of LDS.  All threads call it.
__device__ __forceinline__ void st_wave_smallest(const unsigned long long (&v)[4], int kk, unsigned long long *s_w, unsigned long long *s_res)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long last = 0ull;
    for (int r = 0; r < kk; ++r) {
        unsigned long long best = ~0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((r == 0 || v[u] > last) && v[u] < best) best = v[u];
        last = wave_min_u64(best);                      // ~0 once the wave's keys are used up
        if (lane == 0) s_w[wave * 16 + r] = last;
    }
    __syncthreads();
    if (wave == 0) {
        unsigned long long c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = lane + 64 * u;                // i = wave * kk + r
            c[u] = i < (kStThreads / 64) * kk ? s_w[(i / kk) * 16 + (i % kk)] : ~0ull;
        }
        unsigned long long prev = 0ull;
        for (int r = 0; r < kk; ++r) {
            unsigned long long best = ~0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if ((r == 0 || c[u] > prev) && c[u] < best) best = c[u];
            prev = wave_min_u64(best);
            if (lane == 0) s_res[r] = prev;
        }
    }
    __syncthreads();
}

template <int NV>
__global__ __launch_bounds__(kStThreads) void slice_topk_kernel(SliceArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int MK = p.M * p.Ks, tid = threadIdx.x, lane = tid & 63, k = p.topk;
    const int cap_keys = p.pass_cap + p.k1max;
    float *lds = reinterpret_cast<float *>(smem);
    float *s_q = reinterpret_cast<float *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));
    pq64_t *s_key = reinterpret_cast<pq64_t *>(reinterpret_cast<unsigned char *>(s_q) + (((size_t) p.M * p.Ds * 4 + 15) & ~(size_t) 15));   // [cap_keys]
    unsigned long long *s_buf = s_key + cap_keys;                                                    // [kStBuf]
    unsigned long long *s_out = s_buf + kStBuf;                                                      // [kStRank]
    unsigned int *s_hist = reinterpret_cast<unsigned int *>(s_out + kStRank);                        // [256]
    unsigned int *s_ctl = s_hist + 256;                                                              // [8]
    unsigned long long *s_w = s_buf;                                                                 // [16][16] (wave path: s_buf is free)
    unsigned long long *s_carry = s_out;                                                             // [16]
    const int64_t b = blockIdx.y;
    const int G = (int) gridDim.x, g = (int) blockIdx.x;
    const int64_t n = p.n;
    const int k1 = (int) ((int64_t) k + 1 < n ? (int64_t) k + 1 : n);
    const int64_t per = (n + G - 1) / G;
    const int64_t s_lo = (int64_t) g * per < n ? (int64_t) g * per : n, s_hi = s_lo + per < n ? s_lo + per : n;
    // CPT codes per thread and trip, all their rows in flight before the first lookup; the rows of the very first trip are
    // requested BEFORE the table is built (they do not depend on it): every phase of a block is an exposed memory round trip
    constexpr int CPT = NV == 0 ? 1 : NV <= 2 ? 4 : 2;
    uint4 rows[CPT * NV + 1];
    auto load_rows = [&](int64_t lo, int j0, int cnt) {
        if constexpr (NV > 0) {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int j = j0 + tid + c * kStThreads;
                const int64_t i = lo + (j < cnt ? j : 0);
                const int64_t r = p.remap ? p.remap[i] : i;
                const uint4 *row = reinterpret_cast<const uint4 *>(p.codes + (size_t) r * (NV * 16));
#pragma unroll
                for (int v = 0; v < NV; ++v) rows[c * NV + v] = row[v];
            }
        }
    };
    if (s_lo < s_hi) load_rows(s_lo, 0, (int) (s_hi - s_lo < (int64_t) p.pass_cap ? s_hi - s_lo : (int64_t) p.pass_cap));
    st_build_table(p.queries + (size_t) b * p.M * p.Ds, p.codewords, p.M, p.Ks, p.Ds, p.arch, lds, s_q);
    __syncthreads();
    int carried = 0;
    bool many_any = false;
    const bool wave_path = k1 <= 6;                    // (topk <= 5; measured at N = 1M: -3.4 us at topk = 1, -2 us at 3, +4 us at 10)
    for (int64_t lo = s_lo; lo < s_hi; lo += p.pass_cap) {
        const int cnt = (int) (s_hi - lo < (int64_t) p.pass_cap ? s_hi - lo : (int64_t) p.pass_cap);
        // the pass's keys: four per thread (pass_cap = 4 x 1024), ~0 past the end
        unsigned long long val[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        if constexpr (NV > 0) {
#pragma unroll
            for (int t0 = 0; t0 < 4; t0 += CPT) {
                if (lo != s_lo || t0 != 0) load_rows(lo, t0 * kStThreads, cnt);
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    const int j = (t0 + c) * kStThreads + tid;
                    uint4 w[NV];
#pragma unroll
                    for (int v = 0; v < NV; ++v) w[v] = rows[c * NV + v];
                    if (j < cnt) val[t0 + c] = pq64_make(st_adist<NV>(lds, w, p.Ks), (uint32_t) (lo + j));   // RiiCpp::ADist, m order
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = c * kStThreads + tid;
                if (j < cnt) {
                    const int64_t i = lo + j;
                    const uint8_t *code = p.codes + (size_t) (p.remap ? p.remap[i] : i) * p.M;
                    val[c] = pq64_make(exact_adist(lds, code, p.M, p.Ks), (uint32_t) i);
                }
            }
        }
        const int tot = carried + cnt, kk = k1 < tot ? k1 : tot;
        if (wave_path) {
            // the pass's own smallest by wave-level selection (no histogram, two barriers), then merged with the carried ones
            st_wave_smallest(val, kk < cnt ? kk : cnt, s_w, s_buf + 512);                   // -> s_buf[512 ..)
            const int kn = kk < cnt ? kk : cnt;
            // merge of two sorted runs (carried, new), kk <= 16 outputs: one thread per output position by rank counting
            if (tid < carried + kn) {
                const bool from_old = tid < carried;
                const unsigned long long mine = from_old ? s_carry[tid] : s_buf[512 + tid - carried];
                int rank = 0;
                for (int j = 0; j < carried; ++j) rank += s_carry[j] < mine ? 1 : 0;
                for (int j = 0; j < kn; ++j) rank += s_buf[512 + j] < mine ? 1 : 0;
                if (rank < kk) s_buf[1024 + rank] = mine;
            }
            __syncthreads();
            if (tid < kk) s_carry[tid] = s_buf[1024 + tid];
            carried = kk;
            __syncthreads();
            continue;
        }
        pq64_t *dst = s_key + carried;                 // histogram path: the carried keys stay in front of the pass's keys
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = c * kStThreads + tid;
            if (j < cnt) dst[j] = val[c];
        }
        bool many;
        const unsigned long long *res = st_smallest(s_key, tot, kk, s_buf, s_out, s_hist, s_ctl, &many);   // (barrier first)
        many_any = many_any || many;
        __syncthreads();
        for (int j = tid; j < kk; j += kStThreads) s_key[j] = res[j];
        carried = kk;
        __syncthreads();
    }
    if (wave_path) {                                   // where the tail of the kernel expects the carried keys
        if (tid < carried) s_key[tid] = s_carry[tid];
        __syncthreads();
    }
    {
        unsigned long long *out = p.cand + ((size_t) b * G + g) * p.k1max;
        for (int j = tid; j < k1; j += kStThreads) out[j] = j < carried ? s_key[j] : ~0ull;
    }
    // ---- the last block to arrive merges (release / acquire at device scope around the counter, as in small_topk_kernel) ----
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        s_ctl[6] = atomicAdd(&p.done[b], 1u + (many_any ? 0x10000u : 0u));
        __threadfence();
    }
    __syncthreads();
    const unsigned int arrived = s_ctl[6];
    if ((arrived & 0xffffu) != (unsigned int) (G - 1)) return;
    if (tid == 0) { p.done[b] = 0u; s_ctl[5] = 0u; }
    __syncthreads();
    const bool many_before = (arrived >> 16) != 0u || many_any;
    const unsigned long long *src = p.cand + (size_t) b * G * p.k1max;
    const int total = G * k1;                          // (host: G * k1max <= cap_keys)
    for (int i0 = 0; i0 < total; i0 += 8 * kStThreads) {
        pq64_t e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kStThreads + tid, ic = i < total ? i : total - 1;
            e[u] = __hip_atomic_load(src + (size_t) (ic / k1) * p.k1max + (ic % k1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past this XCD's L2
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kStThreads + tid;
            const bool keep = i < total && e[u] != ~0ull;
            const unsigned long long bal = __ballot(keep);
            if (bal) {
                unsigned int at0 = 0u;
                if (lane == 0) at0 = atomicAdd(&s_ctl[5], (unsigned int) __popcll(bal));
                at0 = (unsigned int) __shfl((int) at0, 0);
                if (keep) s_key[at0 + (unsigned int) __popcll(bal & ((1ull << lane) - 1ull))] = e[u];
            }
        }
    }
    __syncthreads();
    const int tot = (int) s_ctl[5], kk = k1 < tot ? k1 : tot;       // tot >= min(k + 1, n)
    bool many = false;
    const unsigned long long *res;
    if (wave_path) {
        unsigned long long val[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) val[u] = u * kStThreads + tid < tot ? s_key[u * kStThreads + tid] : ~0ull;    // tot <= G k1 <= 4096
        st_wave_smallest(val, kk, s_w, s_buf + 512);
        res = s_buf + 512;
    } else {
        res = st_smallest(s_key, tot, kk, s_buf, s_out, s_hist, s_ctl, &many);
    }
    int tie = (many || many_before) ? 1 : 0;
    if (!many)
        for (int j = tid; j + 1 < kk; j += kStThreads)
            if ((res[j] >> 32) == (res[j + 1] >> 32)) tie = 1;
    tie = __syncthreads_or(tie);
    for (int j = tid; j < k; j += kStThreads) {
        const pq64_t e = res[j < kk ? j : kk - 1];
        const uint32_t idx = pq64_id(e);
        p.out_ids[b * k + j] = p.remap ? p.remap[idx] : (int64_t) idx;
        p.out_dists[b * k + j] = pq64_dist(e);
    }
    if (tid == 0) {
        p.out_tie[b] = tie;                            // 1: the host reruns the call on the general path (std::partial_sort's order) ...
        if (tie && p.flag_list) p.flag_list[atomicAdd(p.nflag, 1)] = (int32_t) b;    // ... or the flag-gated tie kernels behind this launch do
    }
    if (p.host_flag) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every thread's row stores have left the CU before the barrier
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();
            __hip_atomic_store(&p.host_flag[b], p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

constexpr int kSlicePass = 4096;             // codes scored per pass of a block
constexpr int kSliceMaxTopk = 128;
constexpr int64_t kSliceMaxPerBlock = 32768; // codes per block beyond which the general path is the better deal

static size_t slice_topk_smem(int M, int Ks, int Ds, int k1max)
{
    return (((size_t) M * Ks * 4 + 15) & ~(size_t) 15) + (((size_t) M * Ds * 4 + 15) & ~(size_t) 15) + (size_t) (kSlicePass + k1max) * 8 +
           (size_t) (kStBuf + kStRank) * 8 + 256 * 4 + 32;
}
int slice_topk_slices(int64_t n, int64_t B, int topk)
{
    const int64_t by_b = 256 / std::max<int64_t>(B, 1), by_k = kSlicePass / (topk + 1), by_n = (n + kStThreads - 1) / kStThreads;
    return (int) std::max<int64_t>(1, std::min<int64_t>(std::min(by_b, by_k), by_n));
}
bool slice_topk_supported(int M, int Ks, int Ds, int64_t n, int64_t B, int topk)
{
    if (n < 2 || n >= ((int64_t) 1 << 32) || topk < 1 || topk > kSliceMaxTopk || topk > n || B < 1 || B > 8) return false;
    if ((int64_t) M * Ds > 4 * kStThreads || slice_topk_smem(M, Ks, Ds, topk + 1) > (size_t) 160 * 1024 - 512) return false;
    const int G = slice_topk_slices(n, B, topk);
    return (n + G - 1) / G <= kSliceMaxPerBlock;
}
size_t slice_topk_scratch(int64_t n, int64_t B, int topk) { return (size_t) B * slice_topk_slices(n, B, topk) * (topk + 1) * 8; }

hipError_t launch_slice_topk(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_queries, const float *d_codewords, int Ds, int arch,
                             int64_t B, int topk, const int64_t *d_remap, unsigned long long *d_cand, unsigned int *d_done,
                             int64_t *d_out_ids, float *d_out_dists, int32_t *d_out_tie, hipStream_t st, unsigned int *host_flag, unsigned int seq,
                             int32_t *d_flag_list, int *d_nflag)
{
    if (B == 0) return hipSuccess;
    SliceArgs a;
    a.flag_list = d_flag_list; a.nflag = d_nflag;
    a.codes = d_codes; a.n = n; a.M = M; a.Ks = Ks; a.queries = d_queries; a.codewords = d_codewords; a.Ds = Ds; a.arch = arch;
    a.remap = d_remap; a.topk = topk; a.k1max = topk + 1; a.pass_cap = kSlicePass; a.cand = d_cand; a.done = d_done;
    a.out_ids = d_out_ids; a.out_dists = d_out_dists; a.out_tie = d_out_tie; a.host_flag = host_flag; a.seq = seq;
    const size_t smem = slice_topk_smem(M, Ks, Ds, topk + 1);
    void (*kern)(SliceArgs) = slice_topk_kernel<0>;
    if (M == 16) kern = slice_topk_kernel<1>;
    else if (M == 32) kern = slice_topk_kernel<2>;
    else if (M == 48) kern = slice_topk_kernel<3>;
    else if (M == 64) kern = slice_topk_kernel<4>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned) slice_topk_slices(n, B, topk), (unsigned) B), dim3(kStThreads), smem, st, a);
    return hipGetLastError();
}

}  // namespace riiamd
