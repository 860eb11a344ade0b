// comm.hip -- the exchange step of the sharded query path behind the C ABI (round 4): an RCCL communicator (one per process and
// GPU, over xGMI), the packed-record all-gather, and the small kernels around it.  The reference has no multi-device code
// (SURVEY 8e); the parity target is the single-index answer on the concatenated database.
//
// RCCL is bound at RUN time (dlopen): the copy already mapped into the process -- PyTorch-ROCm ships one and a process should hold
// one -- else the system's librccl.so.1.  librii_amd.so itself keeps linking libamdhip64 only; rii_comm_init fails loudly where no
// RCCL can be loaded.  Nothing here falls back to a host exchange.
#include "rii_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>

namespace riiamd {

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                                    // a copy that is already in the process wins
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char *de = dlerror();
        g_rccl.error = std::string("cannot load librccl.so.1 (") + (de ? de : "?") + ")";
        return;
    }
    g_rccl.handle = h;
#define RII_SYM(field, name)                                                        \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));        \
    if (!g_rccl.field) { g_rccl.error = std::string("librccl lacks ") + name; return; }
    RII_SYM(GetUniqueId, "ncclGetUniqueId")
    RII_SYM(CommInitRank, "ncclCommInitRank")
    RII_SYM(CommDestroy, "ncclCommDestroy")
    RII_SYM(AllGather, "ncclAllGather")
    RII_SYM(GetErrorString, "ncclGetErrorString")
#undef RII_SYM
}
}  // namespace

const char *rccl_load_error()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.error.empty() ? nullptr : g_rccl.error.c_str();
}

static_assert(sizeof(ncclUniqueId) == 128, "RII_COMM_ID_BYTES of include/rii_amd.h");

const char *comm_unique_id(void *out128)
{
    if (const char *e = rccl_load_error()) return e;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    __builtin_memcpy(out128, &id, sizeof(id));
    return nullptr;
}

const char *comm_create(const void *id128, int rank, int G, void **out_comm)
{
    if (const char *e = rccl_load_error()) return e;
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&c, G, id, rank);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    *out_comm = c;
    return nullptr;
}

void comm_destroy(void *comm)
{
    if (comm && g_rccl.CommDestroy) (void) g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
}

// every rank's `bytes` bytes at d_send, concatenated in rank order at d_recv; asynchronous on st
const char *comm_all_gather(void *comm, const void *d_send, void *d_recv, size_t bytes, hipStream_t st)
{
    const ncclResult_t r = g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, static_cast<ncclComm_t>(comm), st);
    return r == ncclSuccess ? nullptr : g_rccl.GetErrorString(r);
}

// ---------------------------------------------------------------------------------------------------------------------
// query sharding: rank r answers rows shard_begin(r) .. shard_begin(r + 1) of the batch (sizes differ by at most one row).  Its
// record: [nmax * k] int64 ids, [nmax] int64 counts (inverted index only), [nmax * k] f32 distances, padded to 16 bytes, nmax =
// ceil(B / G).  qshard_unpack_kernel lays the G gathered records out as the [B, k] outputs.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int64_t qs_begin(int64_t B, int G, int r)
{
    const int64_t base = B / G, rem = B % G;
    return r * base + (r < rem ? r : rem);
}
int64_t qshard_begin(int64_t B, int G, int r) { return qs_begin(B, G, r); }
size_t qshard_record_bytes(int64_t B, int G, int k, int counts)
{
    const int64_t nmax = (B + G - 1) / G;
    return ((size_t) nmax * k * 12 + (counts ? (size_t) nmax * 8 : 0) + 15) / 16 * 16;
}

__global__ __launch_bounds__(256) void qshard_unpack_kernel(const unsigned char *__restrict__ gathered, int64_t B, int G, int k, int counts,
                                                            size_t rec, int64_t *__restrict__ out_ids, float *__restrict__ out_dists,
                                                            int64_t *__restrict__ out_counts)
{
    const int64_t nmax = (B + G - 1) / G;
    const int64_t base = B / G, rem = B % G;
    const int64_t total = B * k;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t b = i / k, j = i - b * k;
        // owner of row b: the first `rem` ranks hold base + 1 rows
        int r;
        int64_t lb;
        if (b < rem * (base + 1)) { r = (int) (b / (base + 1)); lb = b - (int64_t) r * (base + 1); }
        else { r = (int) (rem + (b - rem * (base + 1)) / (base > 0 ? base : 1)); lb = b - qs_begin(B, G, r); }
        const unsigned char *p = gathered + rec * (size_t) r;
        out_ids[i] = reinterpret_cast<const int64_t *>(p)[lb * k + j];
        out_dists[i] = reinterpret_cast<const float *>(p + (size_t) nmax * k * 8 + (counts ? (size_t) nmax * 8 : 0))[lb * k + j];
        if (counts && j == 0) out_counts[b] = reinterpret_cast<const int64_t *>(p + (size_t) nmax * k * 8)[lb];
    }
}

hipError_t launch_qshard_unpack(const void *d_gathered, int64_t B, int G, int k, int counts, int64_t *d_out_ids, float *d_out_dists,
                                int64_t *d_out_counts, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    const int64_t total = B * k;
    const int grid = (int) std::min<int64_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(qshard_unpack_kernel, dim3(grid), dim3(256), 0, st, static_cast<const unsigned char *>(d_gathered), B, G, k, counts,
                       qshard_record_bytes(B, G, k, counts), d_out_ids, d_out_dists, d_out_counts);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// database sharding, top-1: the G gathered rows of a query are G (local id, distance) pairs; the answer is the minimum under
// (distance, global id) -- one thread per query instead of merge_topk_kernel's block-wide sort of 64 padded keys.  Record
// layout = merge.hip's ([B] int64 local ids, [B] f32 distances, padded to 16 bytes).
// ---------------------------------------------------------------------------------------------------------------------
struct CommOffsets {
    int64_t off[64];
};
// hdr != 0: the records carry the 16-byte header of merge.hip ({int64 id offset, int32 status}): offsets read from there (any G), and a
// rank that failed poisons every row on every rank (id -2, distance NaN)
__global__ __launch_bounds__(256) void merge_top1_kernel(const unsigned char *__restrict__ gathered, int G, int64_t B, size_t rec,
                                                         CommOffsets offs, int64_t *__restrict__ out_ids, float *__restrict__ out_dists, int hdr)
{
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t best_d = 0xffffffffu;
    int64_t best_id = INT64_MAX;
    int bad = 0;
    for (int g = 0; g < G; ++g) {
        const unsigned char *h = gathered + rec * (size_t) g, *p = h + hdr;
        int64_t id = reinterpret_cast<const int64_t *>(p)[b];
        const uint32_t d = f32_orderable(__float_as_uint(reinterpret_cast<const float *>(p + (size_t) B * 8)[b]));
        if (hdr) bad |= reinterpret_cast<const int32_t *>(h)[2];
        if (!(id < 0 || id >= INT64_MAX / 2)) id += hdr ? *reinterpret_cast<const int64_t *>(h) : offs.off[g];   // padding keys stay what they are
        if (d < best_d || (d == best_d && id < best_id)) { best_d = d; best_id = id; }
    }
    out_ids[b] = bad ? (int64_t) -2 : best_id;
    out_dists[b] = bad ? __uint_as_float(0x7fc00000u) : __uint_as_float(f32_unorderable(best_d));
}
hipError_t launch_merge_top1(const void *d_gathered, int G, int64_t B, const int64_t *id_offsets, int64_t *d_out_ids, float *d_out_dists,
                             hipStream_t st, int hdr)
{
    if (B == 0) return hipSuccess;
    if (G > 64 && !hdr) return hipErrorInvalidValue;
    CommOffsets offs;
    for (int g = 0; g < 64; ++g) offs.off[g] = (id_offsets && g < G) ? id_offsets[g] : 0;
    hipLaunchKernelGGL(merge_top1_kernel, dim3((unsigned) ((B + 255) / 256)), dim3(256), 0, st, static_cast<const unsigned char *>(d_gathered), G, B,
                       merge_record_bytes(B, 1, 0) + (size_t) hdr, offs, d_out_ids, d_out_dists, hdr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// database sharding, top-k: support of the exact-tie replay (tieorder.hip) for the queries the merge flagged.
//   tie_prepare_kernel: for flagged query f = fsel[f]: its query vector gathered into qf[f], and the bound of THIS rank = the
//                       smallest k-th distance any EARLIER shard reported (+inf: none) -- an earlier shard's codes come first in the
//                       reference's index order, so the heap top is at or below it when this shard's first code is visited
//   tie_scatter_kernel: replayed rows into the outputs of the flagged queries whose candidate lists fitted on every rank (count <=
//                       cap); the others keep the (dist, id) rows and are marked in `overflow`
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tie_prepare_kernel(const unsigned char *__restrict__ gathered, size_t rec, int rank, int64_t B, int rows,
                                                          int topk, const int32_t *__restrict__ fsel, int nf, const float *__restrict__ queries,
                                                          int D, float *__restrict__ qf, float *__restrict__ bound)
{   // (gathered = the first rank's rows, rec = the stride of the ranks: the caller has stepped over a record header, if any)
    const int f = blockIdx.x;
    if (f >= nf) return;
    const int64_t b = fsel[f];
    for (int i = threadIdx.x; i < D; i += blockDim.x) qf[(size_t) f * D + i] = queries[(size_t) b * D + i];
    if (threadIdx.x == 0) {
        float bd = INFINITY;
        for (int s = 0; s < rank; ++s) {
            const float *d = reinterpret_cast<const float *>(gathered + rec * (size_t) s + (size_t) B * rows * 8);
            bd = fminf(bd, d[b * rows + (topk - 1)]);                    // +inf when shard s holds fewer than k codes
        }
        bound[f] = bd;
    }
}
hipError_t launch_tie_prepare(const void *d_gathered, int rank, int64_t B, int rows, int topk, const int32_t *d_fsel, int nf,
                              const float *d_queries, int D, float *d_qf, float *d_bound, hipStream_t st, int hdr)
{
    if (nf == 0) return hipSuccess;
    hipLaunchKernelGGL(tie_prepare_kernel, dim3((unsigned) nf), dim3(256), 0, st, static_cast<const unsigned char *>(d_gathered) + hdr,
                       merge_record_bytes(B, rows, 0) + (size_t) hdr, rank, B, rows, topk, d_fsel, nf, d_queries, D, d_qf, d_bound);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void tie_scatter_kernel(const unsigned char *__restrict__ gg, size_t rec2, int G, int nf, int cap, int topk,
                                                          const int32_t *__restrict__ fsel, const int64_t *__restrict__ r_ids,
                                                          const float *__restrict__ r_d, int64_t *__restrict__ out_ids,
                                                          float *__restrict__ out_dists, int32_t *__restrict__ overflow)
{
    const int f = blockIdx.x;
    if (f >= nf) return;
    bool ok = true;
    for (int g = 0; g < G; ++g) ok = ok && (reinterpret_cast<const int32_t *>(gg + rec2 * (size_t) g)[f] <= cap);   // a truncated list cannot be replayed
    const int64_t b = fsel[f];
    if (!ok) {
        if (threadIdx.x == 0 && overflow) overflow[b] = 1;
        return;
    }
    for (int j = threadIdx.x; j < topk; j += blockDim.x) {
        out_ids[b * topk + j] = r_ids[(size_t) f * topk + j];
        out_dists[b * topk + j] = r_d[(size_t) f * topk + j];
    }
}
hipError_t launch_tie_scatter(const void *d_gg, int G, int nf, int cap, int topk, const int32_t *d_fsel, const int64_t *d_r_ids,
                              const float *d_r_d, int64_t *d_out_ids, float *d_out_dists, int32_t *d_overflow, hipStream_t st)
{
    if (nf == 0) return hipSuccess;
    hipLaunchKernelGGL(tie_scatter_kernel, dim3((unsigned) nf), dim3(256), 0, st, static_cast<const unsigned char *>(d_gg),
                       linear_tie_record_bytes(nf, cap), G, nf, cap, topk, d_fsel, d_r_ids, d_r_d, d_out_ids, d_out_dists, d_overflow);
    return hipGetLastError();
}

// the first ncols columns of rows [B, in_stride] -> rows [B, out_stride] (merged k + 1 rows -> their first k; a short local result into
// a padded record); fill_pad_kernel: padding rows of a record (key 2^62: sorts last and is never offset; distance +inf)
__global__ __launch_bounds__(256) void copy_cols_kernel(const int64_t *__restrict__ in_i, const float *__restrict__ in_d, int64_t B, int in_stride,
                                                        int out_stride, int ncols, int64_t *__restrict__ out_i, float *__restrict__ out_d)
{
    const int64_t total = B * ncols;
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t b = i / ncols, j = i - b * ncols;
        out_i[b * out_stride + j] = in_i[b * in_stride + j];
        out_d[b * out_stride + j] = in_d[b * in_stride + j];
    }
}
hipError_t launch_copy_cols(const int64_t *d_in_i, const float *d_in_d, int64_t B, int in_stride, int out_stride, int ncols, int64_t *d_out_i,
                            float *d_out_d, hipStream_t st)
{
    if (B == 0 || ncols == 0) return hipSuccess;
    const int grid = (int) std::min<int64_t>((B * ncols + 255) / 256, 2048);
    hipLaunchKernelGGL(copy_cols_kernel, dim3(grid), dim3(256), 0, st, d_in_i, d_in_d, B, in_stride, out_stride, ncols, d_out_i, d_out_d);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void fill_pad_kernel(int64_t *__restrict__ ids, float *__restrict__ d, int64_t n)
{
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        ids[i] = INT64_MAX / 2;
        d[i] = INFINITY;
    }
}
hipError_t launch_fill_pad(int64_t *d_ids, float *d_d, int64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const int grid = (int) std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(fill_pad_kernel, dim3(grid), dim3(256), 0, st, d_ids, d_d, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// database-sharded inverted index (protocol: ivfshard.hip, include/rii_amd.h): this rank's rows -> the merge record with payload
// ([n] int64 traversal positions, [n] int64 GLOBAL ids, [n] f32 distances; rows the rank does not have: position INT32_MAX, id -1,
// distance +inf), and the finishing touches on the merged rows.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ivf_pack_kernel(const int64_t *__restrict__ ids, const int32_t *__restrict__ pos, const float *__restrict__ d,
                                                       int64_t n, int64_t id_offset, int64_t *__restrict__ rec_pos, int64_t *__restrict__ rec_id,
                                                       float *__restrict__ rec_d)
{
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t id = ids[i];
        rec_pos[i] = (int64_t) pos[i];
        rec_id[i] = id >= 0 ? id + id_offset : id;
        rec_d[i] = d[i];
    }
}
hipError_t launch_ivf_pack(const int64_t *d_ids, const int32_t *d_pos, const float *d_d, int64_t n, int64_t id_offset, void *d_rec, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    unsigned char *r = static_cast<unsigned char *>(d_rec);
    const int grid = (int) std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(ivf_pack_kernel, dim3(grid), dim3(256), 0, st, d_ids, d_pos, d_d, n, id_offset, reinterpret_cast<int64_t *>(r),
                       reinterpret_cast<int64_t *>(r + (size_t) n * 8), reinterpret_cast<float *>(r + (size_t) n * 16));
    return hipGetLastError();
}
// merged rows [B, k1] (payload ids, distances) -> outputs [B, topk]: a query the reference answers with ({}, {}) (count 0) gets
// ids -1 / distances +inf; its tie flag is cleared (nothing to replay), as it is for top-1; a batch a peer poisoned (ids -2,
// distances NaN) gets counts -1
__global__ __launch_bounds__(256) void ivf_finish_kernel(const int64_t *__restrict__ mi, const float *__restrict__ md, const int64_t *__restrict__ cnt,
                                                         int64_t B, int k1, int topk, int64_t *__restrict__ out_ids, float *__restrict__ out_d,
                                                         int64_t *__restrict__ out_cnt, int32_t *__restrict__ tie, int32_t *__restrict__ any)
{
    for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < B * topk; i += (int64_t) gridDim.x * blockDim.x) {
        const int64_t b = i / topk, j = i - b * topk;
        const bool found = cnt[b] > 0;
        const bool poisoned = mi[b * k1] == (int64_t) -2;        // (merge.hip's kPeerFailedId: a peer's header reported a failure)
        out_ids[i] = (found || poisoned) ? mi[b * k1 + j] : (int64_t) -1;
        out_d[i] = (found || poisoned) ? md[b * k1 + j] : INFINITY;
        if (j == 0) {
            // the local count of a poisoned row said "found" -- top-1 callers, who get no status word, read counts -1 there
            out_cnt[b] = poisoned ? (int64_t) -1 : cnt[b];
            const int t = (found && topk > 1) ? tie[b] : 0;
            tie[b] = t;
            if (t) atomicOr(any, 1);
        }
    }
}
hipError_t launch_ivf_finish(const int64_t *d_mi, const float *d_md, const int64_t *d_cnt, int64_t B, int k1, int topk, int64_t *d_out_ids,
                             float *d_out_d, int64_t *d_out_cnt, int32_t *d_tie, int32_t *d_any, hipStream_t st)
{
    if (B == 0) return hipSuccess;
    const int grid = (int) std::min<int64_t>((B * topk + 255) / 256, 2048);
    hipLaunchKernelGGL(ivf_finish_kernel, dim3(grid), dim3(256), 0, st, d_mi, d_md, d_cnt, B, k1, topk, d_out_ids, d_out_d, d_out_cnt, d_tie, d_any);
    return hipGetLastError();
}
// rows fsel[f] of src [.][D] -> dst [nf][D] (the flagged queries' vectors); rows r [nf][k] -> rows fsel[f] of out [.][k]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ fsel, int D, float *__restrict__ dst)
{
    const int f = blockIdx.x;
    const int64_t b = fsel[f];
    for (int i = threadIdx.x; i < D; i += blockDim.x) dst[(size_t) f * D + i] = src[(size_t) b * D + i];
}
hipError_t launch_gather_rows(const float *d_src, const int32_t *d_fsel, int nf, int D, float *d_dst, hipStream_t st)
{
    if (nf == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned) nf), dim3(256), 0, st, d_src, d_fsel, D, d_dst);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const int32_t *__restrict__ fsel, int k, const int64_t *__restrict__ r_i,
                                                           const float *__restrict__ r_d, int64_t *__restrict__ out_i, float *__restrict__ out_d)
{
    const int f = blockIdx.x;
    const int64_t b = fsel[f];
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        out_i[b * k + j] = r_i[(size_t) f * k + j];
        out_d[b * k + j] = r_d[(size_t) f * k + j];
    }
}
hipError_t launch_scatter_rows(const int32_t *d_fsel, int nf, int k, const int64_t *d_r_i, const float *d_r_d, int64_t *d_out_i, float *d_out_d,
                               hipStream_t st)
{
    if (nf == 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned) nf), dim3(256), 0, st, d_fsel, k, d_r_i, d_r_d, d_out_i, d_out_d);
    return hipGetLastError();
}

}  // namespace riiamd
