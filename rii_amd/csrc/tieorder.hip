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
    pq64_t *g_heap;                      // [gridDim.x][topk] heap storage when it does not fit LDS (else unused)
    int heap_in_lds;
    int capl;                            // list capacity (tie_list_cap)
    int indirect;                        // 1: `codes` is the whole database, index i stands for the code remap[i]
    int first;                           // flagged queries [first, nflag) are this kernel's (the chunked path took the others)
};

__device__ __forceinline__ void tie_flush(unsigned long long *list, unsigned int c, pq64_t *heap, long k,
                                          uint32_t *s_thr, unsigned int *s_cnt, int tid, bool wave_heap)
{
    // uniform: c was read between two barriers
    if (c) {
        int nsort = 64;
        while (nsort < (int) c) nsort <<= 1;
        for (int i = tid; i < nsort; i += 256)
            if ((unsigned int) i >= c) list[i] = ~0ull;
        rr_bitonic_sort(list, tid, nsort);                 // ascending (index << 32 | orderable dist): index order
        if (tid < 64 && wave_heap) {                       // wave 0 replays __heap_select (bits/stl_algo.h), 64 entries per round trip
            pq64_t topv = wh_uniform(heap[0]);
            for (unsigned int j0 = 0; j0 < c; j0 += 64) {
                const unsigned int j = j0 + (unsigned int) tid;
                const unsigned long long e = j < c ? list[j] : 0ull;
                const pq64_t v = (e << 32) | (e >> 32);                    // (orderable dist, index)
                unsigned long long m = __ballot(j < c && pq64_less(v, topv));
                while (m) {
                    const int u = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const pq64_t vu = wh_readlane(v, u);
                    if (pq64_less(vu, topv)) {                             // __pop_heap(first, middle, i)
                        topv = wh_adjust_top(heap, (int) k, vu, tid);
                    }
                }
            }
            if (tid == 0) { *s_thr = (uint32_t) (topv >> 32); *s_cnt = 0u; }
        } else if (tid == 0 && !wave_heap) {
            pq64_t topv = heap[0];
            for (unsigned int j0 = 0; j0 < c; j0 += 8) {
                unsigned long long e[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = (j0 + u < c) ? list[j0 + u] : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const pq64_t v = (e[u] << 32) | (e[u] >> 32);
                    if (j0 + u < c && pq64_less(v, topv)) {
                        pq64_adjust_heap(heap, 0, k, v);
                        topv = heap[0];
                    }
                }
            }
            *s_thr = (uint32_t) (topv >> 32);
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
    uint32_t *s_thr = reinterpret_cast<uint32_t *>(list + p.capl);     // orderable bits of the heap top's distance
    unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_thr + 1);
    const long k = p.topk;
    pq64_t *heap = p.heap_in_lds ? reinterpret_cast<pq64_t *>(list + p.capl + 1) : p.g_heap + (size_t) blockIdx.x * k;
    const int tid = threadIdx.x;
    const int nflag = *p.nflag;
    for (int fi = p.first + blockIdx.x; fi < nflag; fi += gridDim.x) {
        __syncthreads();                                   // the previous query's LDS contents are dead from here on
        const int64_t b = p.b0 + p.flag_list[fi];
        {
            const float *src = p.lut + (size_t) (b / p.QT) * MK * p.QT + (b % p.QT);
            for (int i = tid; i < MK; i += 256) lds[i] = src[(size_t) i * p.QT];
        }
        __syncthreads();
        for (long i = tid; i < k; i += 256)                // the first k scores are the initial heap contents
            heap[i] = pq64_make(exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[i] : (int64_t) i) * p.M, p.M, p.Ks), (uint32_t) i);
        if (!p.heap_in_lds) __threadfence_block();
        __syncthreads();
        const bool wave_heap = k <= 2 * 64 * kWhMaxWords && p.heap_in_lds;     // one wave walks the LDS heap (rii_device.h), else one lane
        if (wave_heap && tid < 64) wh_make_heap(heap, (int) k, tid);           // __make_heap, bits/stl_heap.h
        if (!wave_heap && tid == 0) pq64_make_heap(heap, k);
        if (tid == 0) {
            __threadfence_block();
            *s_thr = (uint32_t) (heap[0] >> 32);
            *s_cnt = 0u;
        }
        const int U = p.capl / 512;                        // codes per thread and trip: 4 or 1
        for (int64_t base = k; base < p.n; base += (int64_t) U * 256) {
            __syncthreads();                               // appends of the previous trip are complete ...
            const unsigned int c = *s_cnt;
            __syncthreads();                               // ... and everybody saw the same count before the next ones
            if (c + (unsigned int) U * 256u > (unsigned int) p.capl) tie_flush(list, c, heap, k, s_thr, s_cnt, tid, wave_heap);
            const uint32_t thr = *s_thr;
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * 256 + tid;
                d[u] = (u < U && i < p.n) ? exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[i] : (int64_t) i) * p.M, p.M, p.Ks) : INFINITY;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = base + u * 256 + tid;
                const uint32_t od = f32_orderable(__float_as_uint(d[u]));
                if (u < U && i < p.n && od < thr) list[atomicAdd(s_cnt, 1u)] = ((unsigned long long) (uint32_t) i << 32) | od;
            }
        }
        __syncthreads();
        const unsigned int c = *s_cnt;
        __syncthreads();
        tie_flush(list, c, heap, k, s_thr, s_cnt, tid, wave_heap);
        if (wave_heap && tid < 64) wh_sort_heap(heap, (int) k, tid);           // __sort_heap
        if (!wave_heap && tid == 0) pq64_sort_heap(heap, k);
        if (!p.heap_in_lds) __threadfence_block();
        __syncthreads();
        for (long j = tid; j < k; j += 256) {
            const pq64_t e = heap[j];
            const uint32_t idx = pq64_id(e);
            p.out_ids[b * k + j] = p.remap ? p.remap[idx] : (int64_t) idx;
            p.out_dists[b * k + j] = pq64_dist(e);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Few flagged queries (the usual case: a handful of exact ties in a batch): one block scanning all n codes for a query is
// a ~ms tail behind a sub-ms batch.  The first `fq` flagged queries therefore take three small launches instead:
//   1. tie_chunk_kth_kernel   grid (chunks of 8192 codes, fq): the k-th smallest exact distance of every chunk;
//   2. tie_chunk_emit_kernel  same grid: bound of chunk c = min of the k-th smallest of the chunks before it -- an upper bound
//                             on the heap top while chunk c is visited (each of those chunks alone holds k elements of the
//                             prefix that small) -- and the chunk's codes below the bound are written, in index order, to the
//                             chunk's segment of a per-query list (chunk 0 has no bound: everything, the first k included);
//   3. tie_replay_kernel      one wave per query replays __make_heap / __heap_select / __sort_heap over the segments.
// Same superset argument as above, so the outcome is the full run's.  The exact distances are evaluated by ~n/8192 blocks
// per query instead of one; what stays serial is the heap (about k (1 + ln(n/k)) sifts).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTcChunk = 8192;

struct TcArgs {
    const uint8_t *codes; int64_t n; int M, Ks;
    const float *lut; int QT; int64_t b0;
    const int32_t *flag_list; const int *nflag; const int64_t *remap; int indirect;
    int topk, fq, nchunks;
    uint32_t *kth;                       // [fq][nchunks] orderable bits of the chunk's k-th smallest distance (~0 = fewer than k codes)
    unsigned long long *elist;           // [fq][nchunks * kTcChunk] (index << 32 | orderable dist), ascending index inside a segment
    int32_t *ecount;                     // [fq][nchunks]
    int64_t *out_ids; float *out_dists;
    // database sharding (launch_linear_tie_emit): an upper bound on the heap top from the shards in front of this one, and
    // the compacted candidate list as output instead of the replay
    const float *ext_bound = nullptr;    // [fq] (+inf = none)
    int64_t *em_ids = nullptr; float *em_dists = nullptr; int32_t *em_count = nullptr; int em_cap = 0; int64_t id_offset = 0;
    // round 4, tables above the LDS budget (widetab.hip): the exact distances of query fi were already written by scan_wide_kernel as the
    // key row keyrow[flag_list[fi] * n + i] = (orderable distance << 32 | i); no table is staged
    const unsigned long long *keyrow = nullptr;
    // round 4, the asynchronous few-query path (engine.hip: query_linear_dev, slice_topk_kernel): no table was built for the batch -- a
    // flagged query's block builds it here (RiiCpp::DTable, the arithmetic of lut_build_kernel) instead of a table launch behind EVERY call
    const float *queries = nullptr; const float *codewords = nullptr; int Ds = 0, arch = 0;
    int *nflag_next = nullptr;           // the counter of the NEXT call (two alternate): zeroed by tie_replay_kernel -- no memset per call
};

__device__ __forceinline__ void tc_stage(const TcArgs &p, int64_t b, float *lds, int tid)
{
    if (p.keyrow) return;
    const int MK = p.M * p.Ks;
    if (p.queries) {
        const float *q = p.queries + b * (int64_t) (p.M * p.Ds);
        for (int i = tid; i < MK; i += 256) lds[i] = fvec_l2sqr_any(q + (size_t) (i / p.Ks) * p.Ds, p.codewords + (size_t) i * p.Ds, p.Ds, p.arch);
        return;
    }
    const float *src = p.lut + (size_t) (b / p.QT) * MK * p.QT + (b % p.QT);
    for (int i = tid; i < MK; i += 256) lds[i] = src[(size_t) i * p.QT];
}
// orderable bits of the exact distance of position i for the query whose table (or key row) is number b
__device__ __forceinline__ uint32_t tc_od(const TcArgs &p, const float *lds, int64_t b, int64_t i)
{
    if (p.keyrow) return (uint32_t) (p.keyrow[(size_t) (b - p.b0) * p.n + i] >> 32);
    return f32_orderable(__float_as_uint(exact_adist(lds, p.codes + (size_t) (p.indirect ? (int64_t) p.remap[i] : i) * p.M, p.M, p.Ks)));
}

__global__ __launch_bounds__(256) void tie_chunk_kth_kernel(TcArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int fi = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int nf = *p.nflag < p.fq ? *p.nflag : p.fq;
    if (fi >= nf) return;
    const int MK = p.keyrow ? 0 : p.M * p.Ks;
    float *lds = reinterpret_cast<float *>(smem);
    uint32_t *buf = reinterpret_cast<uint32_t *>(smem + (((size_t) MK * 4 + 15) & ~(size_t) 15));     // [kTcChunk] orderable distances
    const int64_t bq = p.b0 + p.flag_list[fi];
    tc_stage(p, bq, lds, tid);
    __syncthreads();
    const int64_t s = (int64_t) c * kTcChunk;
    const int cnt = (int) ((p.n - s) < kTcChunk ? (p.n - s) : kTcChunk);
    for (int j = tid; j < kTcChunk; j += 256) buf[j] = j < cnt ? tc_od(p, lds, bq, s + j) : 0xffffffffu;
    __syncthreads();
    // k-th smallest of buf[0, cnt), exactly: histograms of the occupied range refined down to single values (cap = 0)
    __shared__ unsigned int s_hist[256];
    __shared__ unsigned int s_ctl[8];
    if (cnt < p.topk) { if (tid == 0) p.kth[(size_t) fi * p.nchunks + c] = 0xffffffffu; return; }
    const uint32_t kth = block_kth_bound([&](int j) { return buf[j]; }, cnt, (uint32_t) p.topk, 0u, s_hist, s_ctl);
    if (tid == 0) p.kth[(size_t) fi * p.nchunks + c] = kth;
}

__global__ __launch_bounds__(256) void tie_chunk_emit_kernel(TcArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int fi = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int nf = *p.nflag < p.fq ? *p.nflag : p.fq;
    if (fi >= nf) return;
    float *lds = reinterpret_cast<float *>(smem);
    __shared__ uint32_t s_bound;
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int64_t bq = p.b0 + p.flag_list[fi];
    tc_stage(p, bq, lds, tid);
    if (tid == 0) { s_bound = 0xffffffffu; s_base = 0; }
    __syncthreads();
    {
        uint32_t b = 0xffffffffu;
        for (int cc = tid; cc < c; cc += 256) {
            const uint32_t v = p.kth[(size_t) fi * p.nchunks + cc];
            b = v < b ? v : b;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t) __shfl_xor((int) b, off);
            b = o < b ? o : b;
        }
        if ((tid & 63) == 0) atomicMin(&s_bound, b);
    }
    __syncthreads();
    if (p.ext_bound && tid == 0) {                         // codes of earlier shards come first in the reference's index order
        const float eb = p.ext_bound[fi];
        if (eb < INFINITY) atomicMin(&s_bound, f32_orderable(__float_as_uint(eb)));
    }
    __syncthreads();
    const uint32_t bound = s_bound;                        // chunk 0: ~0 = everything (a real distance is never ~0: not NaN)
    const int64_t s = (int64_t) c * kTcChunk;
    const int cnt = (int) ((p.n - s) < kTcChunk ? (p.n - s) : kTcChunk);
    unsigned long long *seg = p.elist + ((size_t) fi * p.nchunks + c) * kTcChunk;
    for (int j0 = 0; j0 < cnt; j0 += 256) {                // slabs of 256 consecutive indices: order-preserving compaction
        const int j = j0 + tid;
        uint32_t od = 0xffffffffu;
        if (j < cnt) od = tc_od(p, lds, bq, s + j);
        const bool keep = j < cnt && (bound == 0xffffffffu || od < bound);
        const unsigned long long bal = __ballot(keep);
        const int lane = tid & 63, wave = tid >> 6;
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int w2 = 0; w2 < wave; ++w2) off += s_wave[w2];
        if (keep) seg[off + __popcll(bal & ((1ull << lane) - 1ull))] = ((unsigned long long) (uint32_t) (s + j) << 32) | od;
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) p.ecount[(size_t) fi * p.nchunks + c] = s_base;
}

__global__ __launch_bounds__(64) void tie_replay_kernel(TcArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pq64_t *heap = reinterpret_cast<pq64_t *>(smem);       // [topk]
    const int fi = blockIdx.x, lane = threadIdx.x;
    const int nf = *p.nflag < p.fq ? *p.nflag : p.fq;
    if (p.nflag_next && fi == 0 && lane == 0) *p.nflag_next = 0;      // (its last reader, the previous call's replay, is behind us in stream order)
    if (fi >= nf) return;
    const int k = p.topk;
    const int64_t b = p.b0 + p.flag_list[fi];
    const unsigned long long *list = p.elist + (size_t) fi * p.nchunks * kTcChunk;
    const int32_t *ecount = p.ecount + (size_t) fi * p.nchunks;
    // chunk 0 is unbounded: its segment starts with the indices 0 .. k-1 = the initial heap contents
    for (int i = lane; i < k; i += 64) {
        const unsigned long long e = list[i];
        heap[i] = (e << 32) | (e >> 32);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    wh_make_heap(heap, k, lane);
    pq64_t topv = wh_uniform(heap[0]);
    // round 4: nothing the loop needs waits for a global round trip of its own -- the segment lengths of 64 chunks at a time sit in
    // the lanes (one load per 64 chunks), and the first entries of the NEXT non-empty segment are requested before this segment's sifts
    // (before: a dependent length load and a dependent first load per chunk, ~2 us x 123 chunks of a 1M-code index)
    int cnts = 0;                                             // lane l: ecount[cbase + l]
    int nc = 0;                                               // next chunk whose first entries are in flight / its length
    auto cnt_of = [&](int c) { return __builtin_amdgcn_readlane(cnts, c & 63); };
    cnts = lane < p.nchunks ? ecount[lane] : 0;
    unsigned long long enext = 0ull;
    {
        const int cnt0 = cnt_of(0);
        enext = (k + lane < cnt0) ? list[k + lane] : 0ull;
    }
    for (int c = 0; c < p.nchunks; c = nc) {
        const int cnt = cnt_of(c);
        const unsigned long long *seg = list + (size_t) c * kTcChunk;
        int j0 = c == 0 ? k : 0;
        unsigned long long e = enext;
        // the next chunk with entries (lengths of the following 64-chunk group fetched when the walk crosses into it)
        nc = c + 1;
        while (nc < p.nchunks) {
            if ((nc & 63) == 0) cnts = nc + lane < p.nchunks ? ecount[nc + lane] : 0;
            if (cnt_of(nc) > 0) break;
            ++nc;
        }
        if (nc < p.nchunks) {
            const int cn = cnt_of(nc);
            enext = lane < cn ? list[(size_t) nc * kTcChunk + lane] : 0ull;
        }
        for (; j0 < cnt; j0 += 64) {
            const unsigned long long cur = e;
            if (j0 + 64 + lane < cnt) e = seg[j0 + 64 + lane];          // next round's entries travel under this round's sifts
            const pq64_t v = (cur << 32) | (cur >> 32);
            unsigned long long m = __ballot(j0 + lane < cnt && pq64_less(v, topv));
            while (m) {
                const int u = __builtin_ctzll(m);
                m &= m - 1ull;
                const pq64_t vu = wh_readlane(v, u);
                if (pq64_less(vu, topv)) {                 // __pop_heap(first, middle, i)
                    topv = wh_adjust_top(heap, k, vu, lane);
                }
            }
        }
    }
    wh_sort_heap(heap, k, lane);
    for (int j = lane; j < k; j += 64) {
        const pq64_t e = heap[j];
        const uint32_t idx = pq64_id(e);
        p.out_ids[b * k + j] = p.remap ? p.remap[idx] : (int64_t) idx;
        p.out_dists[b * k + j] = pq64_dist(e);
    }
}

// database sharding: the segments of one query, concatenated in index order, as (global id, distance) rows
__global__ __launch_bounds__(256) void tie_concat_kernel(TcArgs p)
{
    const int fi = blockIdx.x, tid = threadIdx.x;
    const unsigned long long *list = p.elist + (size_t) fi * p.nchunks * kTcChunk;
    const int32_t *ecount = p.ecount + (size_t) fi * p.nchunks;
    int64_t base = 0;
    for (int c = 0; c < p.nchunks; ++c) {
        const int cnt = ecount[c];
        const unsigned long long *seg = list + (size_t) c * kTcChunk;
        for (int j = tid; j < cnt; j += 256) {
            const int64_t o = base + j;
            if (o < p.em_cap) {
                const unsigned long long e = seg[j];
                const uint32_t idx = (uint32_t) (e >> 32);
                p.em_ids[(size_t) fi * p.em_cap + o] = p.id_offset + (p.remap ? p.remap[idx] : (int64_t) idx);
                p.em_dists[(size_t) fi * p.em_cap + o] = __uint_as_float(f32_unorderable((uint32_t) (e & 0xffffffffu)));
            }
        }
        base += cnt;
    }
    if (tid == 0) p.em_count[fi] = (int32_t) (base > 0x7fffffff ? 0x7fffffff : base);
}

// Database-sharded linear search, exact ties (not in the reference): every rank sends, for a flagged query, the
// codes of its shard that can touch the reference's heap (launch_linear_tie_emit), in index order; the ranks' lists one after
// the other ARE a superset of the entering elements in the reference's index order (shards are contiguous id ranges), so one
// wave replays std::partial_sort (src/rii.h:234-235) over them exactly as tie_replay_kernel does over the chunks of one engine.
// Record of a rank: [nf] int32 counts (padded to a multiple of 8 bytes), [nf * cap] int64 global ids, [nf * cap] f32
// distances, padded to 16 bytes.  The heap payload is the entry's place in the gathered records (g * cap + j).
size_t linear_tie_record_bytes(int64_t nf, int cap)
{
    const size_t cb = ((size_t) nf * 4 + 7) / 8 * 8;
    return (cb + (size_t) nf * cap * 12 + 15) / 16 * 16;
}
__global__ __launch_bounds__(64) void linear_shard_replay_kernel(const unsigned char *__restrict__ gathered, int G, int64_t nf, int cap, int topk,
                                                                 int64_t *__restrict__ out_ids, float *__restrict__ out_dists)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pq64_t *heap = reinterpret_cast<pq64_t *>(smem);       // [topk]
    const int64_t f = blockIdx.x;
    const int lane = threadIdx.x, k = topk;
    const size_t cb = ((size_t) nf * 4 + 7) / 8 * 8;
    const size_t rec = (cb + (size_t) nf * cap * 12 + 15) / 16 * 16;
    auto count_of = [&](int g) {
        const int c = reinterpret_cast<const int32_t *>(gathered + rec * g)[f];
        return c < cap ? c : cap;
    };
    auto dist_of = [&](int g, int j) { return reinterpret_cast<const float *>(gathered + rec * g + cb + (size_t) nf * cap * 8)[f * cap + j]; };
    auto id_of = [&](uint32_t pay) {
        const int g = (int) (pay / (uint32_t) cap), j = (int) (pay % (uint32_t) cap);
        return reinterpret_cast<const int64_t *>(gathered + rec * g + cb)[f * cap + j];
    };
    // the first k entries of the sequence (= the indices 0 .. k-1 of the whole database: nothing bounds them) fill the heap
    int g0 = 0, j0 = 0, filled = 0;
    while (filled < k && g0 < G) {
        const int c = count_of(g0);
        const int take = (c - j0) < (k - filled) ? (c - j0) : (k - filled);
        for (int i = lane; i < take; i += 64) heap[filled + i] = pq64_make(dist_of(g0, j0 + i), (uint32_t) (g0 * cap + j0 + i));
        filled += take;
        j0 += take;
        if (j0 >= c) { ++g0; j0 = 0; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (filled == k) {
        wh_make_heap(heap, k, lane);
        pq64_t topv = wh_uniform(heap[0]);
        for (int g = g0; g < G; ++g) {
            const int cnt = count_of(g);
            for (int jj = (g == g0 ? j0 : 0); jj < cnt; jj += 64) {
                const int j = jj + lane;
                const pq64_t v = j < cnt ? pq64_make(dist_of(g, j), (uint32_t) (g * cap + j)) : 0ull;
                unsigned long long m = __ballot(j < cnt && pq64_less(v, topv));
                while (m) {
                    const int u = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const pq64_t vu = wh_readlane(v, u);
                    if (pq64_less(vu, topv)) topv = wh_adjust_top(heap, k, vu, lane);      // __pop_heap(first, middle, i)
                }
            }
        }
        wh_sort_heap(heap, k, lane);
    }
    for (int j = lane; j < k; j += 64) {
        const bool ok = filled == k;
        const pq64_t e = heap[j];
        out_ids[f * k + j] = ok ? id_of(pq64_id(e)) : (int64_t) -1;          // fewer than k entries in all: a caller's error
        out_dists[f * k + j] = ok ? pq64_dist(e) : INFINITY;
    }
}
hipError_t launch_linear_shard_replay(const void *d_gathered, int G, int64_t nf, int cap, int topk, int64_t *d_out_ids,
                                      float *d_out_dists, hipStream_t st)
{
    if (nf == 0) return hipSuccess;
    hipLaunchKernelGGL(linear_shard_replay_kernel, dim3((unsigned) nf), dim3(64), (size_t) topk * 8, st,
                       static_cast<const unsigned char *>(d_gathered), G, nf, cap, topk, d_out_ids, d_out_dists);
    return hipGetLastError();
}

// the chunked path handles topk <= 2 * 64 * kWhMaxWords (the wave-walked heap) on tables that leave room for a chunk of
// distances in LDS; scratch: fq * nchunks * (kTcChunk * 8 + 8) bytes
bool linear_tie_chunked_supported(int M, int Ks, int topk)
{
    return topk <= 2 * 64 * kWhMaxWords && topk <= kTcChunk &&
           (((size_t) M * Ks * 4 + 15) & ~(size_t) 15) + (size_t) kTcChunk * 4 + 64 <= (size_t) 160 * 1024 - 512;
}
bool linear_tie_chunked_topk_ok(int topk) { return topk <= 2 * 64 * kWhMaxWords && topk <= kTcChunk; }      // key-row form: no table in LDS
int64_t linear_tie_chunks(int64_t n) { return (n + kTcChunk - 1) / kTcChunk; }
size_t linear_tie_chunked_scratch(int64_t n, int fq) { return (size_t) fq * (size_t) linear_tie_chunks(n) * ((size_t) kTcChunk * 8 + 8); }

hipError_t launch_linear_tie_chunked(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, int QT, int64_t b0,
                                     const int32_t *d_flag_list, const int *d_nflag, const int64_t *d_remap, int64_t *d_out_ids,
                                     float *d_out_dists, int topk, int fq, void *d_scratch, int indirect, hipStream_t st,
                                     const float *d_queries, const float *d_codewords, int Ds, int arch, int *d_nflag_next)
{
    TcArgs a;
    a.queries = d_queries; a.codewords = d_codewords; a.Ds = Ds; a.arch = arch; a.nflag_next = d_nflag_next;
    a.codes = d_codes; a.n = n; a.M = M; a.Ks = Ks; a.lut = d_lut; a.QT = QT; a.b0 = b0; a.flag_list = d_flag_list;
    a.nflag = d_nflag; a.remap = d_remap; a.indirect = indirect; a.topk = topk; a.fq = fq;
    a.nchunks = (int) linear_tie_chunks(n);
    unsigned char *sc = static_cast<unsigned char *>(d_scratch);
    a.elist = reinterpret_cast<unsigned long long *>(sc);
    a.kth = reinterpret_cast<uint32_t *>(sc + (size_t) fq * a.nchunks * kTcChunk * 8);
    a.ecount = reinterpret_cast<int32_t *>(a.kth + (size_t) fq * a.nchunks);
    a.out_ids = d_out_ids; a.out_dists = d_out_dists;
    const size_t tab = (((size_t) M * Ks * 4 + 15) & ~(size_t) 15);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tie_chunk_kth_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) (tab + kTcChunk * 4));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(tie_chunk_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) tab);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tie_chunk_kth_kernel, dim3(a.nchunks, fq), dim3(256), tab + kTcChunk * 4, st, a);
    hipLaunchKernelGGL(tie_chunk_emit_kernel, dim3(a.nchunks, fq), dim3(256), tab, st, a);
    hipLaunchKernelGGL(tie_replay_kernel, dim3(fq), dim3(64), (size_t) topk * 8, st, a);
    return hipGetLastError();
}

// database sharding: the candidate lists of fq queries (tables in d_lut at b0 .. b0 + fq - 1) over this shard's n codes
hipError_t launch_linear_tie_emit(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, int QT, int64_t b0,
                                  const int32_t *d_flag_list, const int *d_nflag, const int64_t *d_remap, int topk, int fq,
                                  void *d_scratch, int indirect, const float *d_ext_bound, int64_t id_offset, int cap,
                                  int64_t *d_out_ids, float *d_out_dists, int32_t *d_out_count, hipStream_t st,
                                  const unsigned long long *d_keyrow)
{
    TcArgs a;
    a.keyrow = d_keyrow;
    a.codes = d_codes; a.n = n; a.M = M; a.Ks = Ks; a.lut = d_lut; a.QT = QT; a.b0 = b0; a.flag_list = d_flag_list;
    a.nflag = d_nflag; a.remap = d_remap; a.indirect = indirect; a.topk = topk; a.fq = fq;
    a.nchunks = (int) linear_tie_chunks(n);
    unsigned char *sc = static_cast<unsigned char *>(d_scratch);
    a.elist = reinterpret_cast<unsigned long long *>(sc);
    a.kth = reinterpret_cast<uint32_t *>(sc + (size_t) fq * a.nchunks * kTcChunk * 8);
    a.ecount = reinterpret_cast<int32_t *>(a.kth + (size_t) fq * a.nchunks);
    a.out_ids = nullptr; a.out_dists = nullptr;
    a.ext_bound = d_ext_bound; a.em_ids = d_out_ids; a.em_dists = d_out_dists; a.em_count = d_out_count; a.em_cap = cap;
    a.id_offset = id_offset;
    const size_t tab = d_keyrow ? 0 : (((size_t) M * Ks * 4 + 15) & ~(size_t) 15);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tie_chunk_kth_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) (tab + kTcChunk * 4));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(tie_chunk_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) tab);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(tie_chunk_kth_kernel, dim3(a.nchunks, fq), dim3(256), tab + kTcChunk * 4, st, a);
    hipLaunchKernelGGL(tie_chunk_emit_kernel, dim3(a.nchunks, fq), dim3(256), tab, st, a);
    hipLaunchKernelGGL(tie_concat_kernel, dim3(fq), dim3(256), 0, st, a);
    return hipGetLastError();
}

static size_t tie_smem(int M, int Ks, int topk, bool heap_in_lds)
{
    return (((size_t) M * Ks * 4 + 15) & ~(size_t) 15) + (size_t) (tie_list_cap(M, Ks) + 1) * 8 + (heap_in_lds ? (size_t) topk * 8 : 0) + 16;
}
bool linear_tie_heap_in_lds(int M, int Ks, int topk) { return tie_smem(M, Ks, topk, true) <= (size_t) 160 * 1024 - 512; }
bool linear_tie_supported(int M, int Ks) { return tie_smem(M, Ks, 0, false) <= (size_t) 160 * 1024 - 512; }

hipError_t launch_linear_tie(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, int QT, int64_t b0,
                             const int32_t *d_flag_list, const int *d_nflag, const int64_t *d_remap, int64_t *d_out_ids,
                             float *d_out_dists, int topk, int grid, unsigned long long *d_heap, int indirect, int first, hipStream_t st)
{
    TieArgs a;
    a.first = first;
    a.codes = d_codes; a.n = n; a.M = M; a.Ks = Ks; a.lut = d_lut; a.QT = QT; a.b0 = b0; a.flag_list = d_flag_list;
    a.nflag = d_nflag; a.remap = d_remap; a.out_ids = d_out_ids; a.out_dists = d_out_dists; a.topk = topk;
    a.g_heap = d_heap;
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
