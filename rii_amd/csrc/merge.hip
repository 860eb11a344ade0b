// merge.hip -- database sharding: k-way merge of the per-shard top-k rows after the RCCL all-gather (gfx950).
//
// The reference has no multi-device code; the parity target is the single-index answer on the concatenated
// database.  Every rank contributes, per query, its local top-k as (global id, dist); the merged answer is the k
// smallest of the G*k pairs under (dist asc, id asc).  G*k <= 8192 keys per query: one block bitonic-sorts them in LDS.
#include "rii_internal.h"
#include <algorithm>

namespace riiamd {

constexpr int kMergeMaxKeys = 8192;

// record of rank g inside the gathered buffer (what every rank sends): [B*k] int64 keys (ids, or traversal positions), then --
// with a payload -- [B*k] int64 payload (the ids that go with the positions), then [B*k] f32 dists; padded to 16 bytes
__host__ __device__ __forceinline__ size_t mrg_rec_bytes(int64_t Bk, int payload) { return ((size_t) Bk * (payload ? 20 : 12) + 15) / 16 * 16; }
__device__ __forceinline__ const int64_t *mrg_ids(const unsigned char *base, size_t rec, int g) { return reinterpret_cast<const int64_t *>(base + rec * g); }
__device__ __forceinline__ const int64_t *mrg_pay(const unsigned char *base, size_t rec, int g, int64_t Bk) { return reinterpret_cast<const int64_t *>(base + rec * g + (size_t) Bk * 8); }
__device__ __forceinline__ const float *mrg_d(const unsigned char *base, size_t rec, int g, int64_t Bk, int payload) { return reinterpret_cast<const float *>(base + rec * g + (size_t) Bk * (payload ? 16 : 8)); }

// per-rank id offsets (linear search over contiguous id ranges: the ranks send LOCAL ids, the merge adds the shard's first id),
// passed by value with the launch; all zero = the records already hold global keys
constexpr int kMergeMaxRanks = 64;
struct MergeOffsets {
    int64_t off[kMergeMaxRanks];
};

// hdr != 0 (the sharded entry points of engine.hip, round 5): every rank's record is preceded by a 16-byte header {int64 id offset
// of the rank's shard, int32 status, pad} -- the offsets travel WITH the rows (no cached copy on any rank that a re-sharded index
// could leave stale, no collective that only some ranks issue) and so does a rank's failure: a non-zero status in any header
// poisons every row of the batch on EVERY rank (ids kPeerFailedId, distances NaN) and raises bit 1 of *out_any.
constexpr int64_t kPeerFailedId = -2;
__global__ __launch_bounds__(256) void merge_topk_kernel(const unsigned char *__restrict__ gathered_raw, int G, int64_t B, int k, int k_out,
                                                         int payload, int64_t *__restrict__ out_ids, float *__restrict__ out_dists,
                                                         int64_t *__restrict__ out_payload, MergeOffsets offs, int tie_cols,
                                                         int32_t *__restrict__ out_tie, int32_t *__restrict__ out_any, int hdr,
                                                         unsigned long long *__restrict__ gkeys, int64_t b0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int64_t b = blockIdx.x + b0;
    const int tid = threadIdx.x;
    const int n = G * k;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    // (orderable dist << 32 | g * k + j); more than kMergeMaxKeys of them (round 5: any G x k): the same sort over a slice of global
    // scratch -- a block's barrier orders its own global accesses
    unsigned long long *key = gkeys ? gkeys + (size_t) blockIdx.x * (size_t) n2 : reinterpret_cast<unsigned long long *>(smem);
    const int64_t Bk = B * k;
    const size_t rec = mrg_rec_bytes(Bk, payload) + (size_t) hdr;        // rii_merge_record_bytes (+ the header): the stride of the ranks
    const unsigned char *gathered = gathered_raw + hdr;
    if (hdr) {
        int bad = 0;
        for (int g = tid; g < G; g += 256) bad |= reinterpret_cast<const int32_t *>(gathered_raw + rec * g)[2];
        if (__syncthreads_or(bad)) {
            for (int j = tid; j < k_out; j += 256) {
                out_ids[b * k_out + j] = kPeerFailedId;
                out_dists[b * k_out + j] = __uint_as_float(0x7fc00000u);
                if (payload) out_payload[b * k_out + j] = kPeerFailedId;
            }
            if (tid == 0) {
                if (out_tie) out_tie[b] = 0;
                if (out_any) atomicOr(out_any, 2);
            }
            return;
        }
    }
    for (int i = tid; i < n2; i += 256) {
        unsigned long long kk = ~0ull;
        if (i < n) {
            const int g = i / k, j = i - g * k;
            kk = ((unsigned long long) f32_orderable(__float_as_uint(mrg_d(gathered, rec, g, Bk, payload)[b * k + j])) << 32) | (uint32_t) i;
        }
        key[i] = kk;
    }
    auto id_of = [&](unsigned long long kk) -> int64_t {
        const uint32_t s = (uint32_t) (kk & 0xffffffffu);
        if (s >= (uint32_t) n) return INT64_MAX;
        const int g = (int) (s / (uint32_t) k), j = (int) (s - (uint32_t) g * k);
        const int64_t v = mrg_ids(gathered, rec, g)[b * k + j];
        if (v < 0 || v >= INT64_MAX / 2) return v;                      // padding keys (-1 / huge) stay what they are
        return v + (hdr ? *reinterpret_cast<const int64_t *>(gathered_raw + rec * g) : offs.off[g]);
    };
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < n2 / 2; t += 256) {
                const int i = 2 * t - (t & (stride - 1));
                const int j = i + stride;
                const bool up = ((i & size) == 0);
                const unsigned long long x = key[i], y = key[j];
                bool gt;                                   // x sorts after y under (dist, id)
                if ((x >> 32) != (y >> 32)) gt = x > y;
                else gt = id_of(x) > id_of(y);             // exact distance tie: the ids decide (rare)
                if (gt == up) { key[i] = y; key[j] = x; }
            }
        }
    }
    __syncthreads();
    for (int j = tid; j < k_out; j += 256) {
        const unsigned long long kk = key[j];
        out_ids[b * k_out + j] = id_of(kk);
        out_dists[b * k_out + j] = __uint_as_float(f32_unorderable((uint32_t) (kk >> 32)));
        if (payload) {
            const uint32_t s = (uint32_t) (kk & 0xffffffffu);
            int64_t pv = -1;
            if (s < (uint32_t) n) {
                const int g = (int) (s / (uint32_t) k), jj = (int) (s - (uint32_t) g * k);
                pv = mrg_pay(gathered, rec, g, Bk)[b * k + jj];
            }
            out_payload[b * k_out + j] = pv;
        }
    }
    // tie flag of this query: two of the first tie_cols merged distances are bit-equal and finite -- then the reference's
    // order is std::partial_sort's, not (dist, id): the caller replays it (rii_linear_tie_* / rii_ivf_shard_replay_dev)
    if (out_tie) {
        int tie = 0;
        for (int j = tid; j + 1 < tie_cols; j += 256) {
            const uint32_t a = (uint32_t) (key[j] >> 32), c = (uint32_t) (key[j + 1] >> 32);
            if (a == c && c < f32_orderable(0x7f800000u)) tie = 1;
        }
        tie = __syncthreads_or(tie);
        if (tid == 0) {
            out_tie[b] = tie;
            if (tie && out_any) atomicOr(out_any, 1);
        }
    }
}

int merge_topk_max_keys() { return kMergeMaxKeys; }

// queries per launch when the keys live in global scratch (<= 512 MiB of it), and the scratch a merge of B queries needs (0: LDS)
static int64_t merge_group(int G, int64_t B, int k)
{
    size_t n2 = 64;
    while (n2 < (size_t) G * (size_t) k) n2 <<= 1;
    return std::max<int64_t>(1, std::min<int64_t>(B, (int64_t) (((size_t) 512 << 20) / (n2 * 8))));
}
size_t merge_topk_scratch(int G, int64_t B, int k)
{
    if ((int64_t) G * k <= kMergeMaxKeys) return 0;
    size_t n2 = 64;
    while (n2 < (size_t) G * (size_t) k) n2 <<= 1;
    return (size_t) merge_group(G, B, k) * n2 * 8;
}

// Round 6: the database-sharded inverted index's top-1 batch -- merge + finish in ONE launch, a thread per query.  merge_topk_kernel
// (one 256-thread block and a 64-key bitonic sort per query, whatever G x k is) took 10.2 us for 1024 queries x 2 rows, ivf_finish_kernel
// another launch behind it.  Records as above with payload (keys = traversal positions, payload = global ids), k = 2 rows per query
// and rank, 16-byte headers; the winner is the first minimum under (distance, position) -- positions are unique across ranks --;
// `cnt` (> 0 = found: the global walk's verdict, the same on every rank) decides between the row and ({}, {}); a non-zero status in
// any header poisons the batch (ids -2, distances NaN, counts -1, bit 1 of *out_any) exactly as the two kernels did.
__global__ __launch_bounds__(256) void ivf_merge_top1_kernel(const unsigned char *__restrict__ gathered_raw, int G, int64_t B, int hdr,
                                                             const int64_t *__restrict__ cnt, int64_t *__restrict__ out_ids,
                                                             float *__restrict__ out_d, int64_t *__restrict__ out_cnt,
                                                             int32_t *__restrict__ out_tie, int32_t *__restrict__ out_any)
{
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t Bk = B * 2;
    const size_t rec = mrg_rec_bytes(Bk, 1) + (size_t) hdr;
    const unsigned char *gathered = gathered_raw + hdr;
    int bad = 0;
    for (int g = 0; g < G; ++g) bad |= reinterpret_cast<const int32_t *>(gathered_raw + rec * g)[2];
    if (out_tie) out_tie[b] = 0;
    if (bad) {
        out_ids[b] = kPeerFailedId;
        out_d[b] = __uint_as_float(0x7fc00000u);
        out_cnt[b] = -1;
        if (out_any && threadIdx.x == 0) atomicOr(out_any, 2);
        return;
    }
    unsigned long long bestk = ~0ull;                       // (orderable distance, position): positions fit 32 bits (L <= INT32_MAX - 1)
    int64_t bestid = -1;
    for (int g = 0; g < G; ++g) {
        const int64_t *pos = mrg_ids(gathered, rec, g);
        const int64_t *ids = mrg_pay(gathered, rec, g, Bk);
        const float *dd = mrg_d(gathered, rec, g, Bk, 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t ps = pos[b * 2 + j];
            const unsigned long long kk = ((unsigned long long) f32_orderable(__float_as_uint(dd[b * 2 + j])) << 32) | (uint32_t) ps;
            if (ps >= 0 && ps < (int64_t) INT32_MAX && kk < bestk) { bestk = kk; bestid = ids[b * 2 + j]; }
        }
    }
    const bool found = cnt[b] > 0 && bestk != ~0ull;
    out_ids[b] = found ? bestid : (int64_t) -1;
    out_d[b] = found ? __uint_as_float(f32_unorderable((uint32_t) (bestk >> 32))) : INFINITY;
    out_cnt[b] = cnt[b];
}
hipError_t launch_ivf_merge_top1(const void *d_gathered, int G, int64_t B, int hdr, const int64_t *d_cnt, int64_t *d_out_ids, float *d_out_d,
                                 int64_t *d_out_cnt, int32_t *d_out_tie, int32_t *d_out_any, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(ivf_merge_top1_kernel, dim3((unsigned) ((B + 255) / 256)), dim3(256), 0, st, static_cast<const unsigned char *>(d_gathered), G, B, hdr,
                       d_cnt, d_out_ids, d_out_d, d_out_cnt, d_out_tie, d_out_any);
    return hipGetLastError();
}

size_t merge_record_bytes(int64_t B, int k, int payload) { return mrg_rec_bytes(B * k, payload); }

hipError_t launch_merge_topk(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, int64_t *d_out_ids,
                             float *d_out_dists, int64_t *d_out_payload, hipStream_t st, const int64_t *id_offsets, int tie_cols,
                             int32_t *d_out_tie, int32_t *d_out_any, int hdr, void *d_scratch)
{
    if (B == 0) return hipSuccess;
    if (G > kMergeMaxRanks && id_offsets) return hipErrorInvalidValue;
    MergeOffsets offs;
    for (int g = 0; g < kMergeMaxRanks; ++g) offs.off[g] = (id_offsets && g < G) ? id_offsets[g] : 0;
    if ((int64_t) G * k > kMergeMaxKeys) {                   // keys in global scratch (merge_topk_scratch() bytes), group by group
        if (!d_scratch) return hipErrorInvalidValue;
        const int64_t group = merge_group(G, B, k);
        for (int64_t b0 = 0; b0 < B; b0 += group) {
            const int64_t nb = std::min<int64_t>(group, B - b0);
            hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned) nb), dim3(256), 0, st, static_cast<const unsigned char *>(d_gathered), G, B, k, k_out,
                               payload, d_out_ids, d_out_dists, d_out_payload, offs, tie_cols, d_out_tie, d_out_any, hdr,
                               static_cast<unsigned long long *>(d_scratch), b0);
        }
        return hipGetLastError();
    }
    int n2 = 64;
    while (n2 < G * k) n2 <<= 1;
    const size_t smem = (size_t) n2 * 8;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(merge_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned) B), dim3(256), smem, st,
                       static_cast<const unsigned char *>(d_gathered), G, B, k, k_out, payload, d_out_ids, d_out_dists, d_out_payload,
                       offs, tie_cols, d_out_tie, d_out_any, hdr, static_cast<unsigned long long *>(nullptr), (int64_t) 0);
    return hipGetLastError();
}

}  // namespace riiamd
