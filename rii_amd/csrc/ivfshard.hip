// ivfshard.hip -- inverted-index search over a DATABASE-SHARDED index (gfx950).  Not in the reference (it has no
// multi-device code); the parity target is RiiCpp::QueryIvf (src/rii.h:244-326) on the concatenated database.
//
// Rank r of G holds the codes of a contiguous id range and, for every coarse list, the ids of that range (ascending,
// stored locally).  The coarse centres are replicated, so every rank derives the SAME coarse order for a query (the
// std::partial_sort replay of ivf_exact_lds_kernel, ties and unsorted tail included).  The reference's walk is a global,
// sequential rule -- candidates in list order, ids ascending inside a list, stop at exactly L (rii.h:283-305) -- and
// inside one list "ids ascending" means rank 0's part, then rank 1's, ...  With the per-rank list lengths all-gathered
// once per batch (`glen`, query independent: the target-id filter is shared by the batch) every rank computes the same
// global plan -- traversal position of every candidate, the stop position, the "not found" outcome -- and scores exactly
// the candidates it owns:  the element at offset o of global list `no` belongs to the rank with
// before_r[no] <= o < before_r[no] + len_r[no],  before_r = sum of the lengths of the lower ranks.
//
// Output per query: the rank's k+1 best candidates as (dist, traversal position, local id), ascending by (dist, position).
// The merge across ranks under the same key is the reference's answer for top-1 (first minimum in traversal order) and for
// top-k whenever the k+1 smallest distances are pairwise different (rii_amd/dist.py flags the other queries).
#include "rii_internal.h"
#include "rii_device.h"
#include <algorithm>
#include <stdlib.h>

namespace riiamd {

constexpr int kShardMaxNlistLds = 4096;  // coarse (distance, list) pairs in LDS up to here; above: global scratch, heap in LDS
constexpr int kShardMaxL = 8192;         // candidate keys of a query are sorted in LDS

struct ShardArgs {
    const uint8_t *codes; int M, Ks;
    const float *lut;                    // plain [b][M*Ks] tables
    const uint8_t *centers; int nlist;
    const int64_t *pl_off; const int32_t *pl_ids; const int32_t *list_len;       // this rank's (filtered) lists
    const int32_t *glen; int G, rank;    // [G][nlist] lengths of every rank's (filtered) lists
    int topk; int64_t L; int64_t w;
    int rows;                            // output rows per query: topk + 1, or L (every owned candidate: tie replay)
    int64_t *out_ids; float *out_dists; int32_t *out_pos; int32_t *out_nloc; int64_t *out_counts;
    unsigned char *scratch; size_t per_block;      // BIG: [nlist] pq64 + [nlist + 1] int32 per block
    // round 5, ivf_shard_any_kernel: non-NULL = the block builds its query's exact table itself from (queries, codewords) -- no table
    // launch in front of the kernel, no 4 M Ks bytes per query written to and read back from global memory (as ivf_fused_kernel does)
    const float *queries = nullptr; const float *codewords = nullptr; int Ds = 0, arch = 0;
    // round 5, ivf_shard_any_kernel: the codes in POSTING order (row pp = the code of posting pp of pl_ids: the engine's lcodes copy,
    // unfiltered lists only) -- a list's candidates are one contiguous, coalesced run instead of one random 64-byte HBM sector each
    const uint8_t *lcodes = nullptr;
    // round 6, ivf_shard_any_kernel behind shard_coarse_quad_kernel: the batch's coarse picks ([b][kShardPickStride] keys, ascending) and
    // whether they are conclusive; the tables come in through `lut` then
    const unsigned long long *picks = nullptr; const int32_t *pick_ok = nullptr;
    // round 6: the rows written a second time in the exchange record's form (rii_query_ivf_dbsharded_dev: int64 positions | int64 GLOBAL
    // ids | f32 distances -- what ivf_pack_kernel made of the plain outputs in a launch of its own), or NULL
    int64_t *rec_pos = nullptr, *rec_id = nullptr; float *rec_d = nullptr; int64_t id_offset = 0;
    int32_t *zero2 = nullptr;            // two words block 0 clears (the merge's flag words: a memset launch per batch otherwise)
};

// BIG (nlist above kShardMaxNlistLds -- the reference's default sqrt(N) is 11 k lists at a 125 M-code shard): the coarse order and
// the cumulative counts live in global scratch, the heap of the coarse std::partial_sort (w entries) in LDS
// GTAB (round 4): a table above the LDS budget (widetab.hip: M * Ks * 4 bytes > 144 KiB) is read from global memory where it lies
// (plain [b][M * Ks], L2-resident) -- same arithmetic, same order
template <bool BIG, bool GTAB = false>
__global__ __launch_bounds__(256) void ivf_shard_kernel(ShardArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int MK = p.M * p.Ks;
    const int nlist = p.nlist;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    float *lds = reinterpret_cast<float *>(smem);
    const float *tab = GTAB ? p.lut + (size_t) b * MK : lds;
    unsigned char *base = smem + (GTAB ? 0 : (((size_t) MK * 4 + 15) & ~(size_t) 15));
    const bool w_lds = p.w <= kWhSplitMaxHeap;                        // BIG: the heap of the coarse sort in LDS (walked by a wave)
    const int ncoarse = BIG ? (w_lds ? (int) p.w : 0) : nlist;        // entries of the coarse sequence kept in LDS
    pq64_t *s_head = reinterpret_cast<pq64_t *>(base);                // [ncoarse] (coarse distance, list id)
    int32_t *s_cum_lds = reinterpret_cast<int32_t *>(s_head + ncoarse);   // !BIG: [nlist+1] cumulative GLOBAL candidate counts
    int32_t *s_misc = s_cum_lds + (BIG ? 0 : (nlist + 1));            // [4]
    unsigned char *mine = BIG ? p.scratch + p.per_block * blockIdx.x : nullptr;
    pq64_t *s_coarse = BIG ? reinterpret_cast<pq64_t *>(mine) : s_head;                           // the whole order, [nlist]
    int32_t *s_cum = BIG ? reinterpret_cast<int32_t *>(mine + (size_t) nlist * 8) : s_cum_lds;   // [nlist + 1]
    unsigned long long *s_key = reinterpret_cast<unsigned long long *>(
        smem + ((reinterpret_cast<unsigned char *>(s_misc + 4) - smem + 15) & ~(size_t) 15));      // [pow2 >= L]

    if (b == 0 && tid == 0 && p.zero2) { p.zero2[0] = 0; p.zero2[1] = 0; }
    if constexpr (!GTAB) {
        const float *src = p.lut + (size_t) b * MK;
        for (int i = tid; i < MK; i += 256) lds[i] = src[i];
    }
    __syncthreads();
    for (int c = tid; c < nlist; c += 256) {                                          // src/rii.h:262-264
        const pq64_t e = pq64_make(exact_adist(tab, p.centers + (size_t) c * p.M, p.M, p.Ks), (uint32_t) c);
        if (BIG && w_lds && c < (int) p.w) s_head[c] = e; else s_coarse[c] = e;
    }
    __syncthreads();
    if constexpr (BIG) {
        if (w_lds) {
            if (tid < 64) wh_partial_sort_split(s_head, s_coarse + p.w, (int) p.w, nlist, tid);
            __syncthreads();
            for (int c = tid; c < (int) p.w; c += 256) s_coarse[c] = s_head[c];
        } else if (tid == 0) {
            pq64_partial_sort(s_coarse, (long) p.w, (long) nlist);    // deeper heap than the wave code covers: one lane, global memory
        }
        __syncthreads();
    } else {
        if (tid < 64) wh_partial_sort(s_coarse, (int) p.w, nlist, tid);               // src/rii.h:279-280 (wave 0)
    }
    if (tid == 0) {
        long long cnt = 0;
        int nv = 0;
        bool finished = false;
        for (int c = 0; c < nlist; ++c) {                                             // src/rii.h:286-321, global lengths
            const int no = (int) pq64_id(s_coarse[c]);
            long long len = 0;
            for (int g = 0; g < p.G; ++g) len += p.glen[(size_t) g * nlist + no];
            s_cum[c] = (int) cnt;
            if (cnt + len >= p.L) { cnt = p.L; nv = c + 1; finished = true; break; }
            cnt += len;
            if ((long long) (c + 1) == p.w && cnt >= p.topk) { nv = c + 1; finished = true; break; }
        }
        if (!finished) { cnt = 0; nv = 0; }
        s_cum[nv] = (int) cnt;
        s_misc[0] = (int) cnt; s_misc[1] = nv; s_misc[2] = 0;
        p.out_counts[b] = finished ? p.topk : 0;                                      // src/rii.h:324-325 when 0
    }
    __syncthreads();
    const int ncand = s_misc[0], nv = s_misc[1];
    const int k1 = p.rows;
    int n2 = 64;
    while (n2 < ncand) n2 <<= 1;
    for (int pos = tid; pos < n2; pos += 256) {
        unsigned long long key = ~0ull;
        if (pos < ncand) {
            int lo = 0, hi = nv;                                                      // list holding traversal position pos
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_cum[mid] <= pos) lo = mid; else hi = mid;
            }
            const int no = (int) pq64_id(s_coarse[lo]);
            int before = 0;
            for (int g = 0; g < p.rank; ++g) before += p.glen[(size_t) g * nlist + no];
            const int li = pos - s_cum[lo] - before;                                  // index inside this rank's part of the list
            if (li >= 0 && li < p.list_len[no]) {
                const int32_t id = p.pl_ids[p.pl_off[no] + li];
                const float d = exact_adist(tab, p.codes + (size_t) id * p.M, p.M, p.Ks);
                key = ((unsigned long long) f32_orderable(__float_as_uint(d)) << 32) | (uint32_t) pos;
                atomicAdd(&s_misc[2], 1);
            }
        }
        s_key[pos] = key;
    }
    rr_bitonic_sort(s_key, tid, n2);
    const int nloc = s_misc[2] < k1 ? s_misc[2] : k1;
    if (tid == 0) p.out_nloc[b] = nloc;
    for (int j = tid; j < k1; j += 256) {
        int64_t id = -1;
        float d = INFINITY;
        int32_t pos = INT32_MAX;
        if (j < nloc) {
            const unsigned long long key = s_key[j];
            pos = (int32_t) (key & 0xffffffffu);
            d = __uint_as_float(f32_unorderable((uint32_t) (key >> 32)));
            int lo = 0, hi = nv;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_cum[mid] <= pos) lo = mid; else hi = mid;
            }
            const int no = (int) pq64_id(s_coarse[lo]);
            int before = 0;
            for (int g = 0; g < p.rank; ++g) before += p.glen[(size_t) g * nlist + no];
            id = p.pl_ids[p.pl_off[no] + (pos - s_cum[lo] - before)];
        }
        p.out_ids[b * k1 + j] = id;
        p.out_dists[b * k1 + j] = d;
        p.out_pos[b * k1 + j] = pos;
        if (p.rec_pos) {
            p.rec_pos[b * k1 + j] = (int64_t) pos;
            p.rec_id[b * k1 + j] = id >= 0 ? id + p.id_offset : id;
            p.rec_d[b * k1 + j] = d;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Any L (round 5).  The reference's own billion-scale run asks for L = N / nlist = sqrt(N) ~ 31.6 k candidates per query
// (examples/benchmark/run_sift1b.py:105-106, rii/rii.py:143) -- four times what the kernel above sorts in LDS.  Here the coarse order
// and the cumulative counts always live in global scratch (the BIG layout above), the walk visits the visited lists one after the
// other and every thread takes candidates THIS RANK OWNS only (offsets [before_r, before_r + len_r) of a global list clipped to the
// list's share of L: no per-position search), and one of two things happens to a scored candidate:
//   rows = k + 1 (selection): its key (distance, position) joins an LDS buffer of `nbuf` keys if it beats the current bound; a full
//       buffer is sorted, cut back to its k + 1 best and the (k + 1)-th key becomes the bound -- positions are unique, so whatever
//       order the atomics hand out slots in, the sorted prefix is the same.  Needs rows + 512 <= nbuf.
//   rows = L (every owned candidate: exact-tie replay, and the collect-all route of large k): the row goes to slot `position` of the
//       query's output; the slots of candidates other ranks own are padding.  No order, no sort: the replay rebuilds by position.
// ---------------------------------------------------------------------------------------------------------------------
// Several code rows against one table with their LOADS IN FLIGHT TOGETHER (round 5).  exact_adist() fetches its code inside a loop
// over a run-time M, so four calls in a row are four dependent global round trips whatever the unrolling around them; here the rows
// of the wide shapes (M a multiple of 16, <= 64: 16-byte pieces) land in registers first and are scored afterwards -- the same
// sequential fp32 sum over m per row (RiiCpp::ADist, src/rii.h:386-394).
template <int U>
__device__ __forceinline__ void shard_adist_rows(const float *tab, const uint8_t *const (&row)[U], int M, int Ks, float (&out)[U])
{
    if ((M & 15) == 0 && M <= 64) {
        const int MQ = M >> 4;
        uint4 cv[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint4 *cp = reinterpret_cast<const uint4 *>(row[u]);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                if (qd < MQ) cv[u][qd] = cp[qd];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float dist = 0.f;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                if (qd < MQ) {
                    const uint32_t wds[4] = {cv[u][qd].x, cv[u][qd].y, cv[u][qd].z, cv[u][qd].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            dist = __fadd_rn(dist, tab[((qd * 4 + i) * 4 + j) * Ks + ((wds[i] >> (8 * j)) & 0xffu)]);
                }
            }
            out[u] = dist;
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) out[u] = exact_adist(tab, row[u], M, Ks);
    }
}


constexpr int kShardFastW = 7;           // the fast coarse selection covers w <= 7 (w + 1 keys per thread in registers)

// ---------------------------------------------------------------------------------------------------------------------
// shard_coarse_quad_kernel (round 6): the coarse phase of a whole batch in front of ivf_shard_any_kernel, FOUR queries per block.
//
// Inside the one-query block the coarse scores were nlist x M random 4-byte LDS gathers per query (31 of the kernel's 85 us at the
// Deep1B shape: nlist = 8000, M = 16; 65 % of its LDS cycles bank-conflict replays, profiles/r05_deepivf_pmc.json), every block read
// every centre, and every one of the B x nlist coarse keys went out to global scratch (64 MB per launch) for a replay that one query
// in thousands takes.  Here -- ivf_quad_kernel's layout, for any even Ds --
//   tables   four queries' tables interleaved [m][ks][query] in 16-byte rows (thread = (ks, quarter of the subspaces): one codeword
//            load serves four queries; fvec_L2sqr's operations, src/distance.h:117-252), and each query's plain table written to
//            global memory once (M x Ks x 4 bytes, coalesced) for the walk kernel to load instead of rebuilding it;
//   scores   thread = centre: its code is read once per FOUR queries and each of its M lookups is one ds_read_b128 that returns the
//            four queries' entries (sequential fp32 adds over m per query, src/rii.h:375-384);
//   picks    the fast selection of ivf_shard_any_kernel (three smallest keys per thread and query in registers, the w + 1 smallest per
//            wave by DPP minima, one wave per query merges the sixteen waves' picks); `ok` = the w + 1 smallest distances are
//            pairwise different and no thread can have dropped a key that belongs among them -- std::partial_sort's first w entries
//            ARE picks[0 .. w) then (src/rii.h:279-280).  A query without `ok`, or whose walk leaves the first w lists, is scored
//            and replayed by its own block of the walk kernel exactly as before.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kCoarseQ = 4;
constexpr int kCoarseThreads = 1024;
constexpr int kCoarseR = kShardFastW + 1;
static_assert(kCoarseR == kShardPickStride, "the picks' row stride");
constexpr int kCoarseT = 3;              // keys a thread keeps per query

struct CoarseArgs {
    const float *queries, *codewords;
    const uint8_t *centers;
    int M, nlist, w, arch, dbg;
    int64_t B;
    float *lut;                          // [B][M * 256] plain tables (out)
    pq64_t *picks;                       // [B][kCoarseR] the w + 1 smallest (distance, list) keys, ascending (out)
    int32_t *ok;                         // [B] 1 = picks conclusive (out)
};

// s_q: the block's four queries staged in LDS ([4][M * DS], padding queries = the block's first): a thread's query values are
// wave-uniform broadcast reads.  (Fetched straight from global memory they came as VECTOR loads, one dependent round trip per subspace
// -- the table stores may alias them, which rules scalar loads out: four round trips in front of the scores.)
template <int DS, int ARCH, int M>
__device__ __forceinline__ void coarse_quad_tables(const CoarseArgs &p, float4 *__restrict__ lds4, float *__restrict__ s_q, int64_t q0, int nq, int tid)
{
    static_assert(DS % 2 == 0, "8-byte codeword loads");
    constexpr int MK = M * 256, D = M * DS, mper = M / 4, U = DS >= 8 ? 2 : 4;
    const int ks = tid & 255, mg = tid >> 8;
    float *dst[kCoarseQ];
#pragma unroll
    for (int q = 0; q < kCoarseQ; ++q) dst[q] = p.lut + (size_t) (q0 + (q < nq ? q : 0)) * MK + ks;       // (padding: the first query's values again)
    float2 cv[2][U][DS / 2];
    auto request = [&](float2 (&c)[U][DS / 2], int u0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float2 *src = reinterpret_cast<const float2 *>(p.codewords + ((size_t) (mg * mper + u0 + u) * 256 + ks) * DS);
#pragma unroll
            for (int i = 0; i < DS / 2; ++i) c[u][i] = src[i];
        }
    };
    constexpr bool DB = DS < 8;                                  // the next batch's codewords in flight behind this batch's entries
    request(cv[0], 0);
    for (int i = tid; i < kCoarseQ * D; i += kCoarseThreads) {
        const int q = i / D;
        s_q[i] = p.queries[(q0 + (q < nq ? q : 0)) * (int64_t) D + (i - q * D)];
    }
    __syncthreads();
#pragma unroll
    for (int u0 = 0; u0 < mper; u0 += U) {
        const int cur = DB ? (u0 / U) & 1 : 0;
        if (DB && u0 + U < mper) request(cv[cur ^ 1], u0 + U);
        if (!DB && u0) request(cv[0], u0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = mg * mper + u0 + u;
            float y[DS], r[kCoarseQ];
#pragma unroll
            for (int i = 0; i < DS / 2; ++i) { y[2 * i] = cv[cur][u][i].x; y[2 * i + 1] = cv[cur][u][i].y; }
#pragma unroll
            for (int q = 0; q < kCoarseQ; ++q) {
                float x[DS];
#pragma unroll
                for (int i = 0; i < DS / 2; ++i) {
                    const float2 t = *reinterpret_cast<const float2 *>(s_q + q * D + m * DS + 2 * i);
                    x[2 * i] = t.x; x[2 * i + 1] = t.y;
                }
                r[q] = fvec_l2sqr_body(x, y, DS, ARCH);
            }
            lds4[m * 256 + ks] = make_float4(r[0], r[1], r[2], r[3]);
#pragma unroll
            for (int q = 0; q < kCoarseQ; ++q) dst[q][m * 256] = r[q];
        }
    }
}

// MQ = M / 16 (16-byte pieces of a centre's code)
template <int DS, int MQ>
__global__ __launch_bounds__(kCoarseThreads) void shard_coarse_quad_kernel(CoarseArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int M = 16 * MQ, MK = M * 256;
    constexpr int CH = MQ == 1 ? 2 : 1;                                                   // centres per thread in flight (two such sets)
    const int nlist = p.nlist;
    float4 *lds4 = reinterpret_cast<float4 *>(smem);                                      // [MK] rows of four queries' entries
    pq64_t *s_wsel = reinterpret_cast<pq64_t *>(smem + (size_t) MK * 16);                 // [4][16][kCoarseR] the waves' picks
    pq64_t *s_bound = s_wsel + kCoarseQ * 16 * kCoarseR;                                  // [4] the largest pick
    int *s_tie = reinterpret_cast<int *>(s_bound + kCoarseQ);                             // [4]
    int *s_cnt = s_tie + kCoarseQ, *s_over = s_cnt + kCoarseQ;                            // [4] listed keys, [4] list overflow
    float *s_q = reinterpret_cast<float *>(s_over + kCoarseQ);                            // [4][M * DS] the block's queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t q0 = (int64_t) blockIdx.x * kCoarseQ;
    const int nq = (int) (p.B - q0 < kCoarseQ ? p.B - q0 : kCoarseQ);
    const int R = p.w + 1;
    // the codes of this thread's first centres are requested before anything else: their round trip passes behind the table phase
    uint4 cen[2][CH][MQ];
    auto request = [&](uint4 (&dst)[CH][MQ], int c0) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int c = c0 + u * kCoarseThreads;
            const uint4 *cp = reinterpret_cast<const uint4 *>(p.centers + (size_t) (c < nlist ? c : 0) * M);
#pragma unroll
            for (int qd = 0; qd < MQ; ++qd) dst[u][qd] = cp[qd];
        }
    };
    request(cen[0], tid);
    // (up to four floats the three SIMD variants are the same operations: table_rows_regs, rii_device.h)
    if (DS <= 4 || p.arch == RII_SIMD_AVX512) coarse_quad_tables<DS, RII_SIMD_AVX512, M>(p, lds4, s_q, q0, nq, tid);
    else if (p.arch == RII_SIMD_AVX) coarse_quad_tables<DS, RII_SIMD_AVX, M>(p, lds4, s_q, q0, nq, tid);
    else coarse_quad_tables<DS, RII_SIMD_SSE, M>(p, lds4, s_q, q0, nq, tid);
    __syncthreads();
    if (p.dbg == 1) return;
    // ---- scores: thread = centre, four queries per lookup; the next CH centres' codes are in flight while these are scored ----
    pq64_t best[kCoarseQ][kCoarseT];
#pragma unroll
    for (int q = 0; q < kCoarseQ; ++q)
#pragma unroll
        for (int k = 0; k < kCoarseT; ++k) best[q][k] = ~0ull;
    auto score = [&](const uint4 (&cv)[CH][MQ], int c0) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int c = c0 + u * kCoarseThreads;
            if (c >= nlist) break;
            float acc[kCoarseQ] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qd = 0; qd < MQ; ++qd) {
                uint32_t wds[4] = {cv[u][qd].x, cv[u][qd].y, cv[u][qd].z, cv[u][qd].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // eight rows (32 registers) in flight at a time: the code bytes of the next eight lookups "depend" on the sums so far
                    // (left alone the scheduler hoists all 16 MQ reads of a centre -- 64 MQ registers -- and spills)
                    if ((i & 1) == 0 && (qd | i)) {
                        asm volatile("" : "+v"(wds[i]), "+v"(wds[i + 1]) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
                    }
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
            for (int q = 0; q < kCoarseQ; ++q) {
                const pq64_t e = pq64_make(acc[q], (uint32_t) c);
                if (e < best[q][kCoarseT - 1]) {                                           // sorted insertion (keys are distinct: the list id)
                    best[q][kCoarseT - 1] = e;
#pragma unroll
                    for (int k = kCoarseT - 1; k > 0; --k)
                        if (best[q][k] < best[q][k - 1]) { const pq64_t t = best[q][k]; best[q][k] = best[q][k - 1]; best[q][k - 1] = t; }
                }
            }
        }
    };
    for (int c0 = tid; c0 < nlist; c0 += 2 * CH * kCoarseThreads) {
        const int c1 = c0 + CH * kCoarseThreads;
        if (c1 < nlist) request(cen[1], c1);
        score(cen[0], c0);
        if (c1 < nlist) {
            if (c1 + CH * kCoarseThreads < nlist) request(cen[0], c1 + CH * kCoarseThreads);
            score(cen[1], c1);
        }
    }
    if (p.dbg == 2) return;
    pq64_t third[kCoarseQ];
#pragma unroll
    for (int q = 0; q < kCoarseQ; ++q) third[q] = best[q][kCoarseT - 1];                   // (before the pops below)
    // ---- picks.  (First version: every wave extracted its R smallest keys per query -- R lockstep DPP ladders of ~150 instructions on
    // sixteen waves, 5.4 of the kernel's 26 us.)  One ladder per wave: the wave's smallest key per query; the R-th smallest of the
    // sixteen wave minima bounds the R smallest keys of the block from above (the R minima at or below it are R keys already), so
    // only keys at or below that bound -- a handful -- are listed, and one wave per query orders the list. ----
    {
        unsigned long long got[kCoarseQ] = {best[0][0], best[1][0], best[2][0], best[3][0]};
        wave_min_u64_x4(got);
        if (lane < kCoarseQ) s_wsel[lane * 16 + wave] = lane == 0 ? got[0] : lane == 1 ? got[1] : lane == 2 ? got[2] : got[3];
        if (tid < kCoarseQ) s_cnt[tid] = 0;
    }
    __syncthreads();
    if (wave < kCoarseQ) {
        pq64_t cand = lane < 16 ? s_wsel[wave * 16 + lane] : ~0ull;
        pq64_t got = ~0ull;
        for (int r = 0; r < R; ++r) {
            got = wave_min_u64(cand);
            if (cand == got) cand = ~0ull;
        }
        if (lane == 0) s_bound[wave] = got;                                               // (~0 with fewer than R live waves: everything is listed)
    }
    __syncthreads();
    pq64_t *s_list = s_wsel + kCoarseQ * 16;                                              // [4][64]
#pragma unroll
    for (int q = 0; q < kCoarseQ; ++q) {
        const pq64_t bound = s_bound[q];
#pragma unroll
        for (int k = 0; k < kCoarseT; ++k)
            if (best[q][k] != ~0ull && best[q][k] <= bound) {
                const int slot = atomicAdd(&s_cnt[q], 1);
                if (slot < 64) s_list[q * 64 + slot] = best[q][k];
            }
    }
    __syncthreads();
    if (wave < kCoarseQ) {
        const int q = wave;
        const int n = s_cnt[q];
        pq64_t cand = lane < n ? s_list[q * 64 + lane] : ~0ull;
        pq64_t mysel = ~0ull;
        for (int r = 0; r < R; ++r) {
            const pq64_t got = wave_min_u64(cand);
            if (cand == got) cand = ~0ull;
            if (lane == r) mysel = got;
        }
        if (lane == 0) s_over[q] = n > 64 ? 1 : 0;                                        // (more keys at the bound than the list holds: the query is replayed)
        const uint32_t myhi = (uint32_t) (mysel >> 32);
        const uint32_t nxhi = (uint32_t) __shfl_down((int) myhi, 1);
        const bool tied = lane + 1 < R && myhi == nxhi;                                    // exactly tied distances among the w + 1 smallest
        const int tie = __ballot(tied) != 0ull ? 1 : 0;
        if (lane == R - 1) s_bound[q] = mysel;
        if (lane == 0) s_tie[q] = tie;
        if (q < nq && lane < R) p.picks[(size_t) (q0 + q) * kCoarseR + lane] = mysel;
    }
    __syncthreads();
    // a thread whose third key is not above the largest pick may have dropped a fourth that belongs among the picks
    int lostm = 0;
#pragma unroll
    for (int q = 0; q < kCoarseQ; ++q) lostm |= (third[q] != ~0ull && third[q] <= s_bound[q]) ? (1 << q) : 0;
    lostm = __syncthreads_or(lostm);
    if (tid < nq) p.ok[q0 + tid] = (s_tie[tid] == 0 && s_over[tid] == 0 && !((lostm >> tid) & 1)) ? 1 : 0;
}

static size_t shard_coarse_smem(int M, int Ds) { return (size_t) M * 256 * 16 + (size_t) kCoarseQ * 16 * kCoarseR * 8 + kCoarseQ * 8 + 3 * kCoarseQ * 4 + (size_t) kCoarseQ * M * Ds * 4 + 64; }

constexpr int kShardAnyBuf = 8192;       // most keys the selection buffer holds (64 KiB)
constexpr int kShardGroup = 256;         // visited lists whose descriptors are staged per round
constexpr int kShardUnroll = 4;          // candidates a thread scores per round: their ids, then their code rows, are in flight together
constexpr int kShardRound = 256 * kShardUnroll;

// CLDS: the coarse order and the cumulative counts of the query in LDS (nlist <= kShardMaxNlistLds), else in global scratch.
// TOP1: rows == 2 (top-1: the best two owned candidates): every thread keeps its two smallest keys in registers, the block's two
//       smallest come out of two DPP minima per wave and eight keys in LDS -- no buffer, no sort, no atomics.
// (the in-kernel table for an even Ds other than 4: table_rows_regs, rii_device.h)
// PRE (round 6): launched behind shard_coarse_quad_kernel -- tables from `lut`, the coarse order from `picks`
template <bool GTAB, bool CLDS, bool TOP1, bool PRE = false>
__global__ __launch_bounds__(256) void ivf_shard_any_kernel(ShardArgs p, int nbuf, int collect)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int MK = p.M * p.Ks;
    const int nlist = p.nlist;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    if (b == 0 && tid == 0 && p.zero2) { p.zero2[0] = 0; p.zero2[1] = 0; }
    const int dbg = (collect >> 8) & 0xff;                 // measurement only (option "shard_dbg_stop", tools/r5_shard_phases*.sh): return after a phase
    const bool force_replay = (collect >> 16) != 0;        // tests only (option "shard_force_replay"): no fast coarse selection
    collect &= 0xff;
    float *lds = reinterpret_cast<float *>(smem);
    const float *tab = GTAB ? p.lut + (size_t) b * MK : lds;
    unsigned char *base = smem + (GTAB ? 0 : (((size_t) MK * 4 + 15) & ~(size_t) 15));
    const bool w_lds = CLDS || p.w <= kWhSplitMaxHeap;
    const int nhead = CLDS ? nlist : (w_lds ? (int) p.w : 0);
    pq64_t *s_head = reinterpret_cast<pq64_t *>(base);                               // CLDS: the whole order; else the heap of the coarse sort
    int32_t *s_cum_lds = reinterpret_cast<int32_t *>(s_head + nhead);                // CLDS: [nlist + 1]; else [nhead + 2]: the counts of the first w lists
    int32_t *s_misc = s_cum_lds + (CLDS ? nlist + 1 : nhead + 2);                    // [8]: ncand, nv, owned, buffered
    int32_t *s_lpos = s_misc + 8, *s_lown = s_lpos + kShardGroup;                   // staged list descriptors
    int64_t *s_loff = reinterpret_cast<int64_t *>(smem + ((reinterpret_cast<unsigned char *>(s_lown + kShardGroup) - smem + 15) & ~(size_t) 15));
    pq64_t *s_fast = reinterpret_cast<pq64_t *>(s_loff + kShardGroup);                               // !CLDS: [4][w + 1] wave picks, [w] the saved head
    unsigned long long *s_key = s_fast + (CLDS ? 0 : 5 * (kShardFastW + 1));                         // [nbuf] (selection) / [8] (TOP1)
    unsigned char *mine = CLDS ? nullptr : p.scratch + p.per_block * blockIdx.x;
    pq64_t *s_coarse = CLDS ? s_head : reinterpret_cast<pq64_t *>(mine);                            // [nlist] the whole coarse order
    int32_t *s_cum = CLDS ? s_cum_lds : reinterpret_cast<int32_t *>(mine + (size_t) nlist * 8);    // [nlist + 1] cumulative GLOBAL counts

    if constexpr (!GTAB) {
        if (p.queries) {                                                              // RiiCpp::DTable, src/rii.h:361-373 (fvec_L2sqr's order)
            const float *q = p.queries + b * (int64_t) (p.M * p.Ds);
            if (p.Ds == 4 && p.Ks == 256) {                                           // thread = ks: the sub-vector is block-uniform, 16 loads in flight
                const float4 *cw4 = reinterpret_cast<const float4 *>(p.codewords);
                const float4 *q4 = reinterpret_cast<const float4 *>(q);
                for (int m0 = 0; m0 < p.M; m0 += 16) {
                    float4 cv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) cv[u] = (m0 + u < p.M) ? cw4[(m0 + u) * 256 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (m0 + u < p.M) lds[(m0 + u) * 256 + tid] = fvec_l2sqr_ds4v(q4[m0 + u], cv[u]);
                }
            } else if (p.Ks == 256 && p.Ds == 6) {                                    // Deep1B shape (D = 96, M = 16)
                if (p.M == 16) table_rows_regs<6, 8, 16>(lds, q, p.codewords, p.M, p.arch, tid);      // (Deep1B: D = 96, M = 16)
                else table_rows_regs<6, 8>(lds, q, p.codewords, p.M, p.arch, tid);
            } else if (p.Ks == 256 && p.Ds == 8) {
                table_rows_regs<8, 4>(lds, q, p.codewords, p.M, p.arch, tid);
            } else if (p.Ks == 256 && p.Ds == 2) {
                table_rows_regs<2, 16>(lds, q, p.codewords, p.M, p.arch, tid);
            } else {
                for (int m = 0; m < p.M; ++m) {
                    const float *qm = q + (size_t) m * p.Ds;
                    const float *cm = p.codewords + (size_t) m * p.Ks * p.Ds;
                    for (int ks = tid; ks < p.Ks; ks += 256) lds[m * p.Ks + ks] = fvec_l2sqr_any(qm, cm + (size_t) ks * p.Ds, p.Ds, p.arch);
                }
            }
        } else {
            const float *src = p.lut + (size_t) b * MK;
            if ((MK & 3) == 0) {
                const float4 *s4 = reinterpret_cast<const float4 *>(src);
                float4 *d4 = reinterpret_cast<float4 *>(lds);
                for (int i = tid; i < (MK >> 2); i += 256) d4[i] = s4[i];
            } else {
                for (int i = tid; i < MK; i += 256) lds[i] = src[i];
            }
        }
    }
    __syncthreads();
    if (dbg == 1) return;
    // FAST coarse selection (round 5; order in global scratch, w <= kShardFastW): while the centres are scored every thread keeps the
    // w + 1 smallest keys it has seen (sorted, in registers); afterwards each wave extracts the w + 1 smallest of its lanes' lists with
    // DPP minima and wave 0 merges the four waves' picks -- the w + 1 smallest (distance, list) keys of the query in ~3 us instead
    // of the library's heap replayed over thousands of lists (~26 us at 8000 lists).  That IS std::partial_sort's result whenever those
    // w + 1 distances are pairwise different; exactly tied distances among them, or a walk that has to continue past list w (the
    // unsorted tail, whose arrangement only the replay knows), fall back to the exact replay -- ivf_fused_kernel's rule, without a
    // second launch: the original sequence is still in place (head saved, tail untouched).
    constexpr int kFastR = kShardFastW + 1;
    // round 6: the batch's coarse phase ran in shard_coarse_quad_kernel (four queries per block, one 16-byte LDS read per centre lookup):
    // this block starts from that kernel's picks unless they were inconclusive -- no coarse scores, no coarse keys written anywhere
    static_assert(!PRE || (!CLDS && !GTAB), "the pre-pass serves the order-in-scratch, table-in-LDS form");
    bool pre_ok = PRE && !force_replay && p.pick_ok[b] != 0;                           // (block-uniform)
    const bool fast_ok = !PRE && !CLDS && w_lds && p.w <= kShardFastW && nlist > (int) p.w + 64 && !force_replay;
    const int R = (int) p.w + 1;
    const int nlds = CLDS ? nlist : nhead;
    // the order's first w entries (the sorted heap) and the counts up to list w are read from LDS wherever the order itself lives in
    // global scratch: the walk leaves them only for stale lists (tail walk)
    auto order_at = [&](int c) -> pq64_t { return (CLDS || c < nlds) ? s_head[c] : s_coarse[c]; };
    auto cum_at = [&](int c) -> int { return (CLDS || c <= nlds) ? s_cum_lds[c] : s_cum[c]; };
    auto cum_set = [&](int c, int v) { if (CLDS || c <= nlds) s_cum_lds[c] = v; if (!CLDS) s_cum[c] = v; };
    bool fast_used = false, scored = false;
    // scores every centre (the sequence std::partial_sort works on: head in LDS, tail in scratch) and, without a pre-pass, picks;
    // true = the measurement stop after this phase
    auto score_and_pick = [&]() -> bool {
        // (three keys per thread, not w + 1: with 256 threads three of the w + 1 smallest keys of a query share a thread in 0.01 - 0.1 % of
        //  the queries; a thread whose THIRD key is within the picks may have dropped a fourth -- detected, and such a query is replayed.
        //  Eight sorted slots per thread cost the scoring loop 10 us of insertions at 8000 lists.)
        constexpr int kFastT = 3;
        pq64_t best[kFastT];
#pragma unroll
        for (int k = 0; k < kFastT; ++k) best[k] = ~0ull;
        if constexpr (PRE) {                                                       // (rare: one centre at a time keeps this form's registers low)
            for (int c = tid; c < nlist; c += 256) {
                const pq64_t e = pq64_make(exact_adist(tab, p.centers + (size_t) c * p.M, p.M, p.Ks), (uint32_t) c);
                if (w_lds && c < (int) p.w) s_head[c] = e; else s_coarse[c] = e;
            }
        } else
        for (int c0 = tid; c0 < nlist; c0 += 4 * 256) {                           // src/rii.h:262-264; four centres' codes in flight per thread
            float dv[4];
            const uint8_t *crow[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 256;
                crow[u] = p.centers + (size_t) (c < nlist ? c : 0) * p.M;
            }
            shard_adist_rows<4>(tab, crow, p.M, p.Ks, dv);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 256;
                if (c < nlist) {
                    const pq64_t e = pq64_make(dv[u], (uint32_t) c);
                    if (!CLDS && w_lds && c < (int) p.w) s_head[c] = e; else s_coarse[c] = e;
                    if (fast_ok && e < best[kFastT - 1]) {                         // sorted insertion (keys are distinct: the list id)
                        best[kFastT - 1] = e;
#pragma unroll
                        for (int k = kFastT - 1; k > 0; --k)
                            if (best[k] < best[k - 1]) { const pq64_t t = best[k]; best[k] = best[k - 1]; best[k - 1] = t; }
                    }
                }
            }
        }
        __syncthreads();
        if (dbg == 2) return true;
        // FAST coarse selection (round 5; order in global scratch, w <= kShardFastW, no pre-pass): every thread kept its three smallest
        // keys while the centres were scored; each wave extracts the w + 1 smallest of its lanes' lists with DPP minima and wave 0
        // merges the four waves' picks -- the w + 1 smallest (distance, list) keys of the query in ~3 us instead of the library's heap
        // replayed over thousands of lists (~26 us at 8000 lists).  That IS std::partial_sort's result whenever those w + 1 distances
        // are pairwise different; exactly tied distances among them, or a walk that has to continue past list w (the unsorted tail,
        // whose arrangement only the replay knows), fall back to the exact replay -- ivf_fused_kernel's rule, without a second
        // launch: the original sequence is still in place (head saved, tail untouched).
        if (fast_ok) {
            const int lane = tid & 63, wave = tid >> 6;
            const pq64_t third = best[kFastT - 1];                                 // (before the pops below)
            for (int r = 0; r < R; ++r) {                                          // level 1: this wave's R smallest
                const pq64_t got = wave_min_u64(best[0]);
                if (best[0] == got) {                                              // (one lane: it hands the key over and moves up its list)
#pragma unroll
                    for (int k = 0; k + 1 < kFastT; ++k) best[k] = best[k + 1];
                    best[kFastT - 1] = ~0ull;
                }
                if (lane == 0) s_fast[wave * kFastR + r] = got;
            }
            __syncthreads();
            if (wave == 0) {                                                       // level 2: the block's R smallest, ascending; lane r keeps pick r
                pq64_t cand = lane < 4 * R ? s_fast[(lane / R) * kFastR + (lane % R)] : ~0ull;
                pq64_t mysel = ~0ull;
                for (int r = 0; r < R; ++r) {
                    const pq64_t got = wave_min_u64(cand);
                    if (cand == got) cand = ~0ull;
                    if (lane == r) mysel = got;
                }
                const uint32_t myhi = (uint32_t) (mysel >> 32);
                const uint32_t nxhi = (uint32_t) __shfl_down((int) myhi, 1);
                const bool tied = lane + 1 < R && myhi == nxhi;                     // exactly tied distances among the w + 1 smallest
                const int tie = __ballot(tied) != 0ull ? 1 : 0;
                if (lane == R - 1) s_fast[4 * kFastR + kFastR - 1] = mysel;         // the largest pick: the bound of the check below
                if (lane == 0) s_misc[6] = tie;
                if (lane < (int) p.w) s_fast[lane] = mysel;                         // (the waves' picks are dead: the block's picks in their place)
            }
            __syncthreads();
            // a thread whose third key is not above the largest pick may have dropped a fourth that belongs among the picks: replay
            const int lost = __syncthreads_or(third != ~0ull && third <= s_fast[4 * kFastR + kFastR - 1]);
            fast_used = s_misc[6] == 0 && !lost;
            if (fast_used && tid < (int) p.w) {
                const pq64_t pick = s_fast[tid];
                s_fast[4 * kFastR + tid] = s_head[tid];                             // the sequence's own first w entries: kept for a replay
                s_head[tid] = pick;
            }
            __syncthreads();
        }
        return false;
    };
    if constexpr (!PRE) {
        if (score_and_pick()) return;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if constexpr (PRE) {
            if (!pre_ok && !scored) {
                scored = true;
                if (score_and_pick()) return;
            } else if (dbg == 2) return;
        }
        if (PRE && pre_ok) {                                                           // std::partial_sort's first w entries, from the pre-pass
            if (tid < (int) p.w) s_head[tid] = p.picks[(size_t) b * kFastR + tid];
            fast_used = true;
            __syncthreads();
        }
        if (!fast_used) {                                                              // the library's algorithm, move for move
            if constexpr (CLDS) {
                if (tid < 64) wh_partial_sort(s_coarse, (int) p.w, nlist, tid);        // src/rii.h:279-280 (wave 0)
            } else if (w_lds) {
                if (tid < 64) wh_partial_sort_split(s_head, s_coarse + p.w, (int) p.w, nlist, tid);      // heap in LDS, tail in global scratch
                __syncthreads();
            } else if (tid == 0) {
                pq64_partial_sort(s_coarse, (long) p.w, (long) nlist);
            }
        }
        if (!CLDS && w_lds)
            for (int c = tid; c < (int) p.w; c += 256) s_coarse[c] = s_head[c];
        __syncthreads();
        if (dbg == 3) return;
        if (tid == 0) {
            long long cnt = 0;
            int nv = 0;
            bool finished = false, redo = false;
            for (int c = 0; c < nlist; ++c) {                                         // src/rii.h:286-321, global lengths
                if (fast_used && c >= (int) p.w) { redo = true; break; }              // past list w: only the replay knows the tail's order
                const int no = (int) pq64_id(order_at(c));
                long long len = 0;
                for (int g = 0; g < p.G; ++g) len += p.glen[(size_t) g * nlist + no];
                cum_set(c, (int) cnt);
                if (cnt + len >= p.L) { cnt = p.L; nv = c + 1; finished = true; break; }
                cnt += len;
                if ((long long) (c + 1) == p.w && cnt >= p.topk) { nv = c + 1; finished = true; break; }
            }
            if (!finished) { cnt = 0; nv = 0; }
            cum_set(nv, (int) cnt);
            s_misc[0] = (int) cnt; s_misc[1] = nv; s_misc[2] = 0; s_misc[3] = 0; s_misc[7] = redo ? 1 : 0;
            p.out_counts[b] = finished ? p.topk : 0;                                  // src/rii.h:324-325 when 0
        }
        __syncthreads();
        if (!s_misc[7]) break;
        // the walk left the first w lists.  In-kernel picks: the sequence's own head comes back and the library's sort is replayed on
        // it; pre-pass picks: the sequence was never written -- score the centres now, then replay
        if (pre_ok) pre_ok = false;
        else if (tid < (int) p.w) s_head[tid] = s_fast[4 * kFastR + tid];
        fast_used = false;
        __syncthreads();
    }
    const int rows = p.rows;
    if (collect)                                                                       // padding everywhere first; owned rows overwrite it
        for (int j = tid; j < rows; j += 256) {
            p.out_ids[b * rows + j] = -1;
            p.out_dists[b * rows + j] = INFINITY;
            p.out_pos[b * rows + j] = INT32_MAX;
        }
    __syncthreads();
    if (dbg == 4) return;
    const int nv = s_misc[1];
    unsigned long long thr = ~0ull;                                                   // selection: keys at or above it cannot make the cut
    unsigned long long b0 = ~0ull, b1 = ~0ull;                                        // TOP1: this thread's two smallest keys
    int owned = 0;
    auto flush = [&]() {                                                              // all threads; barriers inside
        const int n = s_misc[3];
        int n2 = 64;
        while (n2 < n) n2 <<= 1;
        for (int i = n + tid; i < n2; i += 256) s_key[i] = ~0ull;
        rr_bitonic_sort(s_key, tid, n2);
        const int kept = n < rows ? n : rows;
        if (kept == rows) thr = s_key[rows - 1];
        __syncthreads();
        if (tid == 0) s_misc[3] = kept;
        __syncthreads();
    };
    for (int c0 = 0; c0 < nv; c0 += kShardGroup) {
        __syncthreads();
        if (tid < kShardGroup) {                                                      // this rank's part of visited list c0 + tid
            int lpos = 0, own = 0;
            int64_t off = 0;
            const int c = c0 + tid;
            if (c < nv) {
                const int no = (int) pq64_id(order_at(c));
                const int cum = cum_at(c), take = cum_at(c + 1) - cum;               // the list's share of the L candidates
                int before = 0;
                for (int g = 0; g < p.rank; ++g) before += p.glen[(size_t) g * nlist + no];
                const int mylen = p.list_len[no];
                own = take - before < mylen ? take - before : mylen;
                own = own > 0 ? own : 0;
                lpos = cum + before;
                off = p.pl_off[no];
            }
            s_lpos[tid] = lpos; s_lown[tid] = own; s_loff[tid] = off;
        }
        __syncthreads();
        const int ng = nv - c0 < kShardGroup ? nv - c0 : kShardGroup;
        for (int l = 0; l < ng; ++l) {
            const int own = s_lown[l];
            if (own == 0) continue;
            const int lpos = s_lpos[l];
            const int32_t *ids = p.pl_ids + s_loff[l];
            const uint8_t *lrows = p.lcodes ? p.lcodes + (size_t) s_loff[l] * p.M : nullptr;
            if constexpr (!GTAB) {
                // round 6: posting-order rows of the common shapes (M = 16 / 32, Ks = 256) -- the NEXT round's rows are requested before
                // this round's are scored: the rows stream from HBM (a 1 GB shard), and with one round in flight per block its ~2 us
                // round trip stood in front of every 1024 candidates
                if (lrows && !collect && p.Ks == 256 && (p.M == 16 || p.M == 32)) {
                    auto run = [&](auto mq_tag) {
                        constexpr int MQ = decltype(mq_tag)::value, MM = 16 * MQ;
                        uint4 rv[2][kShardUnroll][MQ];
                        auto req = [&](uint4 (&d)[kShardUnroll][MQ], int base_li) {
#pragma unroll
                            for (int u = 0; u < kShardUnroll; ++u) {
                                const int li = base_li + u * 256 + tid;
                                const uint4 *cp = reinterpret_cast<const uint4 *>(lrows + (size_t) (li < own ? li : 0) * MM);
#pragma unroll
                                for (int qd = 0; qd < MQ; ++qd) d[u][qd] = cp[qd];
                            }
                        };
                        auto score = [&](const uint4 (&d)[kShardUnroll][MQ], int base_li) {
                            if (!TOP1 && __syncthreads_or(s_misc[3] + 2 * kShardRound > nbuf)) flush();       // (see the general loop below)
                            float prev = 0.f;
                            unsigned long long x0 = b0, x1 = b1;          // (value copies: updated through the captured references the pair is demoted to scratch)
                            int nown = 0;
#pragma unroll
                            for (int u = 0; u < kShardUnroll; ++u) {
                                float dist = 0.f;
#pragma unroll
                                for (int qd = 0; qd < MQ; ++qd) {
                                    uint32_t wds[4] = {d[u][qd].x, d[u][qd].y, d[u][qd].z, d[u][qd].w};
                                    // one row piece (16 lookups) in flight at a time: its code bytes "depend" on the sums so far (left alone
                                    // the scheduler hoists every read of the round -- 128 MQ registers -- and the kernel loses two blocks per CU)
                                    asm volatile("" : "+v"(wds[0]), "+v"(wds[1]), "+v"(wds[2]), "+v"(wds[3]) : "v"(prev), "v"(dist));
#pragma unroll
                                    for (int i = 0; i < 4; ++i)
#pragma unroll
                                        for (int j = 0; j < 4; ++j)
                                            dist = __fadd_rn(dist, lds[((qd * 4 + i) * 4 + j) * 256 + ((wds[i] >> (8 * j)) & 0xffu)]);
                                }
                                prev = dist;
                                const int li = base_li + u * 256 + tid;
                                if (li >= own) continue;
                                ++nown;
                                const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(dist)) << 32) | (uint32_t) (lpos + li);
                                if constexpr (TOP1) {
                                    const bool lt0 = key < x0, lt1 = key < x1;
                                    x1 = lt0 ? x0 : (lt1 ? key : x1);
                                    x0 = lt0 ? key : x0;
                                } else {
                                    if (key < thr) s_key[atomicAdd(&s_misc[3], 1)] = key;
                                }
                            }
                            b0 = x0; b1 = x1; owned += nown;
                        };
                        req(rv[0], 0);
                        for (int base_li = 0; base_li < own; base_li += 2 * kShardRound) {
                            const int nxt = base_li + kShardRound;
                            if (nxt < own) req(rv[1], nxt);
                            score(rv[0], base_li);
                            if (nxt < own) {
                                if (nxt + kShardRound < own) req(rv[0], nxt + kShardRound);
                                score(rv[1], nxt);
                            }
                        }
                    };
                    if (p.M == 16) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 2>{});
                    continue;
                }
            }
            for (int base_li = 0; base_li < own; base_li += kShardRound) {
                // a thread reads the fill behind its own insertions of the round before, not behind everyone's: up to one round
                // short of the truth, hence two rounds of slack; the OR makes the decision the block's
                if (!TOP1 && !collect && __syncthreads_or(s_misc[3] + 2 * kShardRound > nbuf)) flush();
                int32_t idv[kShardUnroll];
#pragma unroll
                for (int u = 0; u < kShardUnroll; ++u) {                              // the ids of the round first (posting-order rows: only
                    const int li = base_li + u * 256 + tid;                            //  the collect form needs them at all) ...
                    idv[u] = li < own ? ((lrows && !collect) ? 0 : ids[li]) : -1;
                }
                float dv[kShardUnroll];
                const uint8_t *row[kShardUnroll];
#pragma unroll
                for (int u = 0; u < kShardUnroll; ++u) {                              // ... then their rows (independent loads in flight)
                    const int li = base_li + u * 256 + tid;
                    row[u] = idv[u] < 0 ? p.codes : (lrows ? lrows + (size_t) li * p.M : p.codes + (size_t) idv[u] * p.M);
                }
                shard_adist_rows<kShardUnroll>(tab, row, p.M, p.Ks, dv);
#pragma unroll
                for (int u = 0; u < kShardUnroll; ++u) {
                    if (idv[u] < 0) continue;
                    const int pos = lpos + base_li + u * 256 + tid;
                    ++owned;
                    if (collect) {
                        p.out_ids[b * rows + pos] = idv[u];
                        p.out_dists[b * rows + pos] = dv[u];
                        p.out_pos[b * rows + pos] = pos;
                    } else {
                        const unsigned long long key = ((unsigned long long) f32_orderable(__float_as_uint(dv[u])) << 32) | (uint32_t) pos;
                        if constexpr (TOP1) {
                            if (key < b1) {
                                if (key < b0) { b1 = b0; b0 = key; } else b1 = key;
                            }
                        } else {
                            if (key < thr) s_key[atomicAdd(&s_misc[3], 1)] = key;
                        }
                    }
                }
            }
        }
    }
    atomicAdd(&s_misc[2], owned);
    __syncthreads();
    if (dbg == 5) return;
    const int nown = s_misc[2];
    if (collect) {
        if (tid == 0) p.out_nloc[b] = nown < rows ? nown : rows;
        return;
    }
    if constexpr (TOP1) {
        // the wave's smallest key, then -- the lane that held it falls back on its second -- the wave's second smallest
        const unsigned long long m0 = wave_min_u64(b0);
        const unsigned long long m1 = wave_min_u64(b0 == m0 ? b1 : b0);              // (keys are unique: one lane holds m0)
        if ((tid & 63) == 0) { s_key[2 * (tid >> 6)] = m0; s_key[2 * (tid >> 6) + 1] = m1; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long k0 = ~0ull, k1 = ~0ull;
            for (int i = 0; i < 8; ++i) {
                const unsigned long long k = s_key[i];
                if (k < k1) { if (k < k0) { k1 = k0; k0 = k; } else k1 = k; }
            }
            s_key[0] = k0; s_key[1] = k1;
            s_misc[3] = nown < 2 ? nown : 2;
        }
        __syncthreads();
    } else {
        flush();
    }
    const int nloc = s_misc[3];                                                        // = min(owned, rows)
    if (tid == 0) p.out_nloc[b] = nloc;
    for (int j = tid; j < rows; j += 256) {
        int64_t id = -1;
        float d = INFINITY;
        int32_t pos = INT32_MAX;
        if (j < nloc) {
            const unsigned long long key = s_key[j];
            pos = (int32_t) (key & 0xffffffffu);
            d = __uint_as_float(f32_unorderable((uint32_t) (key >> 32)));
            int lo = 0, hi = nv;                                                      // the list holding traversal position pos
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (cum_at(mid) <= pos) lo = mid; else hi = mid;
            }
            const int no = (int) pq64_id(order_at(lo));
            int before = 0;
            for (int g = 0; g < p.rank; ++g) before += p.glen[(size_t) g * nlist + no];
            id = p.pl_ids[p.pl_off[no] + (pos - cum_at(lo) - before)];
        }
        p.out_ids[b * rows + j] = id;
        p.out_dists[b * rows + j] = d;
        p.out_pos[b * rows + j] = pos;
        if (p.rec_pos) {
            p.rec_pos[b * rows + j] = (int64_t) pos;
            p.rec_id[b * rows + j] = id >= 0 ? id + p.id_offset : id;
            p.rec_d[b * rows + j] = d;
        }
    }
}

// Tie replay for the database-sharded inverted index: when two of the merged k+1 best distances are bit-equal the reference's
// answer is what std::partial_sort (src/rii.h:312-313) makes of the WHOLE candidate sequence in traversal order.  Every rank
// then sends all the candidates it owns for that query (rows = L) and every rank rebuilds the sequence by position and
// replays the library's algorithm on it (one wave, rii_device.h: wh_partial_sort).  Record layout = rii_merge_topk_dev's
// with payload: [nf*rows] int64 positions (INT32_MAX = none), [nf*rows] int64 global ids, [nf*rows] f32 distances.
__global__ __launch_bounds__(256) void shard_replay_kernel(const unsigned char *__restrict__ gathered, int G, int64_t nf, int rows,
                                                           int topk, int64_t *__restrict__ out_ids, float *__restrict__ out_dists)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pq64_t *seq = reinterpret_cast<pq64_t *>(smem);                     // [rows] (dist, position), in traversal order
    int64_t *sid = reinterpret_cast<int64_t *>(seq + rows);             // [rows] global id of the candidate at a position
    __shared__ int s_n;
    const int64_t f = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t n = nf * rows;
    const size_t rec = ((size_t) n * 20 + 15) / 16 * 16;
    if (tid == 0) s_n = 0;
    __syncthreads();
    int mine = 0;
    for (int g = 0; g < G; ++g) {
        const unsigned char *base = gathered + rec * g;
        const int64_t *pos = reinterpret_cast<const int64_t *>(base);
        const int64_t *gid = reinterpret_cast<const int64_t *>(base + (size_t) n * 8);
        const float *dd = reinterpret_cast<const float *>(base + (size_t) n * 16);
        for (int j = tid; j < rows; j += 256) {
            const int64_t ps = pos[f * rows + j];
            if (ps >= 0 && ps < rows) {
                seq[ps] = pq64_make(dd[f * rows + j], (uint32_t) ps);
                sid[ps] = gid[f * rows + j];
                ++mine;
            }
        }
    }
    atomicAdd(&s_n, mine);
    __syncthreads();
    const int ncand = s_n;                                  // positions are dense: every candidate is owned by exactly one rank
    if (tid < 64) wh_partial_sort(seq, topk, ncand, tid);
    __syncthreads();
    for (int j = tid; j < topk; j += 256) {
        const pq64_t e = seq[j];
        out_ids[f * topk + j] = sid[pq64_id(e)];
        out_dists[f * topk + j] = pq64_dist(e);
    }
}

// The same replay for sequences that do not fit LDS (rows > 8192, round 5): the rebuilt sequence and the ids live in global scratch
// (16 bytes per row and query), the heap of std::partial_sort in LDS and walked by a wave while topk <= 1024 (wh_partial_sort_split:
// a tail entry is read once, at its own step), by one lane over global memory above that (the single engine's last resort as well:
// ivf_select_kernel).  `middle` = min(topk, candidates): a query the reference answers with ({}, {}) has no candidates at all.
__global__ __launch_bounds__(256) void shard_replay_any_kernel(const unsigned char *__restrict__ gathered, int G, int64_t nf, int rows,
                                                               int topk, unsigned char *__restrict__ scratch, int64_t *__restrict__ out_ids,
                                                               float *__restrict__ out_dists)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pq64_t *head = reinterpret_cast<pq64_t *>(smem);                    // [topk] when topk <= kWhSplitMaxHeap
    __shared__ int s_n;
    const int64_t f = blockIdx.x;
    const int tid = threadIdx.x;
    pq64_t *seq = reinterpret_cast<pq64_t *>(scratch + (size_t) f * rows * 16);
    int64_t *sid = reinterpret_cast<int64_t *>(seq + rows);
    const int64_t n = nf * rows;
    const size_t rec = ((size_t) n * 20 + 15) / 16 * 16;
    if (tid == 0) s_n = 0;
    __syncthreads();
    int mine = 0;
    for (int g = 0; g < G; ++g) {
        const unsigned char *base = gathered + rec * g;
        const int64_t *pos = reinterpret_cast<const int64_t *>(base);
        const int64_t *gid = reinterpret_cast<const int64_t *>(base + (size_t) n * 8);
        const float *dd = reinterpret_cast<const float *>(base + (size_t) n * 16);
        for (int j = tid; j < rows; j += 256) {
            const int64_t ps = pos[f * rows + j];
            if (ps >= 0 && ps < rows) {
                seq[ps] = pq64_make(dd[f * rows + j], (uint32_t) ps);
                sid[ps] = gid[f * rows + j];
                ++mine;
            }
        }
    }
    atomicAdd(&s_n, mine);
    __syncthreads();
    const int ncand = s_n;                                  // positions are dense: every candidate is owned by exactly one rank
    const int middle = topk < ncand ? topk : ncand;
    const bool in_lds = topk <= kWhSplitMaxHeap;
    if (in_lds) {
        for (int j = tid; j < middle; j += 256) head[j] = seq[j];
        __syncthreads();
        if (tid < 64) wh_partial_sort_split(head, seq + middle, middle, ncand, tid);
    } else if (tid == 0) {
        pq64_partial_sort(seq, (long) middle, (long) ncand);
    }
    __syncthreads();
    for (int j = tid; j < topk; j += 256) {
        int64_t id = -1;
        float d = INFINITY;
        if (j < middle) {
            const pq64_t e = in_lds ? head[j] : seq[j];
            id = sid[pq64_id(e)];
            d = pq64_dist(e);
        }
        out_ids[f * topk + j] = id;
        out_dists[f * topk + j] = d;
    }
}

size_t shard_replay_scratch(int64_t nf, int rows) { return rows > kShardMaxL ? (size_t) nf * (size_t) rows * 16 : 0; }

hipError_t launch_shard_replay(const void *d_gathered, int G, int64_t nf, int rows, int topk, int64_t *d_out_ids,
                               float *d_out_dists, void *d_scratch, hipStream_t st)
{
    if (nf == 0) return hipSuccess;
    if (rows > kShardMaxL) {                                // sequences in global scratch (shard_replay_scratch() bytes)
        if (!d_scratch) return hipErrorInvalidValue;
        const size_t smem = topk <= kWhSplitMaxHeap ? (size_t) topk * 8 : 0;
        hipLaunchKernelGGL(shard_replay_any_kernel, dim3((unsigned) nf), dim3(256), smem, st, static_cast<const unsigned char *>(d_gathered),
                           G, nf, rows, topk, static_cast<unsigned char *>(d_scratch), d_out_ids, d_out_dists);
        return hipGetLastError();
    }
    const size_t smem = (size_t) rows * 16;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(shard_replay_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(shard_replay_kernel, dim3((unsigned) nf), dim3(256), smem, st, static_cast<const unsigned char *>(d_gathered),
                       G, nf, rows, topk, d_out_ids, d_out_dists);
    return hipGetLastError();
}

static bool shard_big(int nlist) { return nlist > kShardMaxNlistLds; }
static bool shard_gtab(int M, int Ks) { return lut_tile_for(M, Ks) == 0; }       // no whole table fits LDS (rii_internal.h)
static size_t shard_smem(int M, int Ks, int nlist, int64_t L, int64_t w)
{
    size_t n2 = 64;
    while ((int64_t) n2 < L) n2 <<= 1;
    const size_t coarse = shard_big(nlist) ? (size_t) (w <= kWhSplitMaxHeap ? w : 0) * 8 : (size_t) nlist * 8 + (size_t) (nlist + 1) * 4;
    return (shard_gtab(M, Ks) ? 0 : (((size_t) M * Ks * 4 + 15) & ~(size_t) 15)) + coarse + 16 + 16 + n2 * 8;
}
// the kernel with every working set of a query in LDS (L <= 8192 candidate keys sorted at once)
static bool shard_lds_ok(int M, int Ks, int nlist, int64_t L, int64_t w)
{
    if (L > kShardMaxL) return false;
    return shard_smem(M, Ks, nlist, L, w) <= (size_t) 160 * 1024 - 512;
}
// ivf_shard_any_kernel: LDS without the selection buffer, and the buffer that fits next to it (a power of two)
static size_t shard_any_misc() { return 8 * 4 + 2 * kShardGroup * 4 + 16 + kShardGroup * 8; }
static size_t shard_tab_bytes(int M, int Ks) { return shard_gtab(M, Ks) ? 0 : (((size_t) M * Ks * 4 + 15) & ~(size_t) 15); }
// the coarse order of the query in LDS: up to kShardMaxNlistLds lists, and only next to a table that leaves room for it and a
// 1024-key buffer
static bool shard_any_clds(int M, int Ks, int nlist)
{
    // ... and only while FOUR blocks still fit a CU (40 KiB each): a 1024-query batch is then one round of blocks, not two -- above
    // that the order goes to global scratch (its tail is read eight slices ahead: two round trips for 1024 lists)
    return !shard_big(nlist) &&
           shard_tab_bytes(M, Ks) + (size_t) nlist * 8 + (size_t) (nlist + 1) * 4 + shard_any_misc() + 64 <= (size_t) 40 * 1024 - 128;
}
static size_t shard_any_fixed(int M, int Ks, int nlist, int64_t w)
{
    const size_t nh = (size_t) (w <= kWhSplitMaxHeap ? w : 0);
    const size_t coarse = shard_any_clds(M, Ks, nlist) ? (size_t) nlist * 8 + (size_t) (nlist + 1) * 4 : nh * 8 + (nh + 2) * 4 + (size_t) 5 * (kShardFastW + 1) * 8;
    return shard_tab_bytes(M, Ks) + coarse + shard_any_misc();
}
static int shard_any_nbuf(int M, int Ks, int nlist, int64_t w)
{
    const size_t avail = (size_t) 160 * 1024 - 512 - shard_any_fixed(M, Ks, nlist, w);
    int nbuf = kShardAnyBuf;
    while (nbuf > 64 && (size_t) nbuf * 8 > avail) nbuf >>= 1;
    return nbuf;
}
static int shard_any_max_rows(int M, int Ks, int nlist, int64_t w)
{
    const int r = shard_any_nbuf(M, Ks, nlist, w) - 2 * kShardRound;                 // two rounds of slack (see the kernel)
    return r > 2 ? r : 2;                                                             // (rows = 2: the register path, no buffer)
}
// rows per query a launch can select (rows = k + 1 form); anything up to L is served by the collect form (rows >= L)
int ivf_shard_max_select_rows(int M, int Ks, int nlist, int64_t L, int64_t w)
{
    if (shard_lds_ok(M, Ks, nlist, L, w)) return kShardMaxL + 1;
    return shard_any_max_rows(M, Ks, nlist, w);
}
bool ivf_shard_supported(int M, int Ks, int nlist, int64_t L, int64_t w, int rows)
{
    // rows >= L (every owned candidate) is always served: by the 8192-key kernel while it holds them, else by the collect form
    if (shard_lds_ok(M, Ks, nlist, L, w)) return rows <= kShardMaxL + 1 || (int64_t) rows >= L;      // (rows = k + 1 with k = L = 8192)
    return rows <= shard_any_max_rows(M, Ks, nlist, w) || (int64_t) rows >= L;
}
// which kernel: top-1 (rows = 2) always takes the register path of ivf_shard_any_kernel (sorting L keys to keep two of them is what
// the 8192-key kernel would do); other row counts the 8192-key kernel while everything fits, else the selection buffer / collect form
static bool shard_use_any(int M, int Ks, int nlist, int64_t L, int64_t w, int rows)
{
    if (rows == 2 && L > 2 && shard_any_fixed(M, Ks, nlist, w) + 64 <= (size_t) 160 * 1024 - 512) return true;
    if (!shard_lds_ok(M, Ks, nlist, L, w)) return true;
    if (rows > kShardMaxL + 1) return true;                  // (rows >= L beyond the 8192-key kernel's rows -> the collect form)
    // everything the selection buffer serves: the any-L kernel has the faster candidate loop (rows in flight together, posting-order
    // rows) and the fast coarse selection; the 8192-key kernel keeps the row counts beyond it (k + 1 up to 8193 at L <= 8192)
    return rows <= shard_any_max_rows(M, Ks, nlist, w) && shard_any_fixed(M, Ks, nlist, w) + (size_t) shard_any_nbuf(M, Ks, nlist, w) * 8 <= (size_t) 160 * 1024 - 512;
}
// bytes of global scratch per query of a launch (coarse order + cumulative counts), 0: everything fits LDS
size_t ivf_shard_scratch_per_query(int M, int Ks, int nlist, int64_t L, int64_t w)
{
    return (shard_big(nlist) || !shard_any_clds(M, Ks, nlist)) ? ((size_t) nlist * 12 + 4 + 63) / 64 * 64 : 0;
}

// true: launch_ivf_shard will run the kernel that builds its tables itself when handed (queries, codewords) instead of d_lut
bool ivf_shard_builds_tables(int M, int Ks, int nlist, int64_t L, int64_t w, int rows)
{
    return !shard_gtab(M, Ks) && shard_use_any(M, Ks, nlist, L, w, rows);
}
// round 6: the coarse pre-pass in front of ivf_shard_any_kernel (shard_coarse_quad_kernel).  Applies where that kernel would build its own
// table and keep the coarse order in global scratch, for the shapes the four-query table fits LDS.
bool shard_coarse_supported(int M, int Ks, int Ds, int nlist, int64_t L, int64_t w, int rows)
{
    return Ks == 256 && (M == 16 || M == 32) && (Ds == 2 || Ds == 4 || Ds == 6 || Ds == 8) && w <= kShardFastW && nlist > (int) w + 64 &&
           !shard_gtab(M, Ks) && shard_use_any(M, Ks, nlist, L, w, rows) && !shard_any_clds(M, Ks, nlist) &&
           shard_coarse_smem(M, Ds) <= (size_t) 160 * 1024 - 512;
}
hipError_t launch_shard_coarse(const float *d_queries, const float *d_codewords, const uint8_t *d_centers, int M, int Ds, int arch, int nlist,
                               int64_t w, int64_t B, float *d_lut, unsigned long long *d_picks, int32_t *d_pick_ok, hipStream_t st, int debug)
{
    if (B == 0) return hipSuccess;
    CoarseArgs a;
    a.dbg = debug;
    a.queries = d_queries; a.codewords = d_codewords; a.centers = d_centers; a.M = M; a.nlist = nlist; a.w = (int) w; a.arch = arch; a.B = B;
    a.lut = d_lut; a.picks = d_picks; a.ok = d_pick_ok;
    auto kern = M == 16 ? (Ds == 2 ? shard_coarse_quad_kernel<2, 1> : Ds == 4 ? shard_coarse_quad_kernel<4, 1> : Ds == 6 ? shard_coarse_quad_kernel<6, 1> : shard_coarse_quad_kernel<8, 1>)
                        : (Ds == 2 ? shard_coarse_quad_kernel<2, 2> : Ds == 4 ? shard_coarse_quad_kernel<4, 2> : Ds == 6 ? shard_coarse_quad_kernel<6, 2> : shard_coarse_quad_kernel<8, 2>);
    const size_t smem = shard_coarse_smem(M, Ds);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned) ((B + kCoarseQ - 1) / kCoarseQ)), dim3(kCoarseThreads), smem, st, a);
    return hipGetLastError();
}
hipError_t launch_ivf_shard(const uint8_t *d_codes, int M, int Ks, const float *d_lut, const uint8_t *d_centers, int nlist,
                            const int64_t *d_pl_off, const int32_t *d_pl_ids, const int32_t *d_list_len, const int32_t *d_glen,
                            int G, int rank, int64_t B, int topk, int64_t L, int64_t w, int rows, int64_t *d_out_ids, float *d_out_dists,
                            int32_t *d_out_pos, int32_t *d_out_nloc, int64_t *d_out_counts, void *d_scratch, hipStream_t st,
                            const float *d_queries, const float *d_codewords, int Ds, int arch, const uint8_t *d_lcodes, int debug,
                            const unsigned long long *d_picks, const int32_t *d_pick_ok, const ShardPack *pack)
{
    if (B == 0) return hipSuccess;
    ShardArgs a;
    a.picks = d_picks; a.pick_ok = d_pick_ok;
    if (pack && pack->rec_pos) { a.rec_pos = pack->rec_pos; a.rec_id = pack->rec_id; a.rec_d = pack->rec_d; a.id_offset = pack->id_offset; }
    if (pack) a.zero2 = pack->zero2;
    a.queries = d_queries; a.codewords = d_codewords; a.Ds = Ds; a.arch = arch; a.lcodes = d_lcodes;
    a.codes = d_codes; a.M = M; a.Ks = Ks; a.lut = d_lut; a.centers = d_centers; a.nlist = nlist; a.pl_off = d_pl_off;
    a.pl_ids = d_pl_ids; a.list_len = d_list_len; a.glen = d_glen; a.G = G; a.rank = rank; a.topk = topk; a.L = L; a.w = w; a.rows = rows;
    a.out_ids = d_out_ids; a.out_dists = d_out_dists; a.out_pos = d_out_pos; a.out_nloc = d_out_nloc; a.out_counts = d_out_counts;
    a.scratch = static_cast<unsigned char *>(d_scratch); a.per_block = ivf_shard_scratch_per_query(M, Ks, nlist, L, w);
    if (shard_use_any(M, Ks, nlist, L, w, rows)) {          // any L: register path (top-1), selection buffer, or collect form
        const bool top1 = rows == 2 && L > 2;
        const int nbuf = shard_any_nbuf(M, Ks, nlist, w);
        const int collect = (!top1 && rows > shard_any_max_rows(M, Ks, nlist, w)) ? 1 : 0;
        if (collect && (int64_t) rows < L) return hipErrorInvalidValue;
        if (collect && a.rec_pos) return hipErrorInvalidValue;       // (the packed form is the selected-rows form's)
        const size_t smem = shard_any_fixed(M, Ks, nlist, w) + (top1 ? 64 : (collect ? 0 : (size_t) nbuf * 8));
        const bool gt = shard_gtab(M, Ks), cl = shard_any_clds(M, Ks, nlist);
        const bool pre = d_picks && d_pick_ok && d_lut && !gt && !cl;
        if (d_picks && !pre) return hipErrorInvalidValue;
        auto kern = top1 ? (gt ? (cl ? ivf_shard_any_kernel<true, true, true> : ivf_shard_any_kernel<true, false, true>)
                               : (cl ? ivf_shard_any_kernel<false, true, true> : pre ? ivf_shard_any_kernel<false, false, true, true> : ivf_shard_any_kernel<false, false, true>))
                         : (gt ? (cl ? ivf_shard_any_kernel<true, true, false> : ivf_shard_any_kernel<true, false, false>)
                               : (cl ? ivf_shard_any_kernel<false, true, false> : pre ? ivf_shard_any_kernel<false, false, false, true> : ivf_shard_any_kernel<false, false, false>));
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
        if (e != hipSuccess) return e;
        // debug (engine options "shard_dbg_stop" | "shard_force_replay" << 8; 0 in production): bits 8..15 of the kernel's word = return after
        // that phase (measurement), bit 16 = no fast coarse selection, every query takes the exact replay (tests)
        hipLaunchKernelGGL(kern, dim3((unsigned) B), dim3(256), smem, st, a, nbuf, collect | ((debug & 0xffff) << 8));
        return hipGetLastError();
    }
    const size_t smem = shard_smem(M, Ks, nlist, L, w);
    auto kern = shard_gtab(M, Ks) ? (shard_big(nlist) ? ivf_shard_kernel<true, true> : ivf_shard_kernel<false, true>)
                                  : (shard_big(nlist) ? ivf_shard_kernel<true> : ivf_shard_kernel<false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned) B), dim3(256), smem, st, a);
    return hipGetLastError();
}

}  // namespace riiamd
