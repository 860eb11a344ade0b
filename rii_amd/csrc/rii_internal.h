// rii_internal.h -- shared declarations between the HIP kernels (kernels.hip, sortsel.hip) and the host
// engine (engine.cpp).  gfx950 / CDNA4 only.  Nothing here is part of the public C ABI (include/rii_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>

namespace riiamd {

// Timing of the kernels that dominate a query step: the engine arms a pair of events (engine.hip: ScopedTimer) and the
// launcher hands them to the dispatch itself (hipExtLaunchKernelGGL: start / stop timestamps of THIS dispatch).  An event
// recorded into the stream is a barrier packet of its own; two of them around every scan cost ~8 % of a 0.4 ms step.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local LaunchEvents g_launch_events;
template <typename K, typename A>
inline void launch_timed(K kern, dim3 grid, dim3 block, size_t smem, hipStream_t st, const A &arg)
{
    if (g_launch_events.start && g_launch_events.stop) {
        hipExtLaunchKernelGGL(kern, grid, block, (uint32_t) smem, st, g_launch_events.start, g_launch_events.stop, 0, arg);
        g_launch_events = LaunchEvents();
    } else {
        hipLaunchKernelGGL(kern, grid, block, smem, st, arg);
    }
}

// ---------------------------------------------------------------------------------------------------
// Device data layouts (all in HBM, owned by the engine)
//   codes      : u8  [N][M] row-major -- identical to the reference's flattened_codes_ (src/rii.h:80), so
//                a posting-list hit is ONE contiguous M-byte gather and a linear scan is a pure stream.
//   codewords  : f32 [M][Ks][Ds]
//   lut        : f32 tile-interleaved [ceil(B/QT)][M][Ks][QT]: the QT queries of one scan tile sit side by
//                side so one ds_read_b128 (QT=4) serves four queries' table entries for one (m, code byte).
//   symtab     : f32 [M][Ks][Ks]  (PQk-means symmetric tables, src/pqkmeans.cpp:23-34)
//   centers    : u8  [nlist][M]
//   pl_off/ids : CSR posting lists (ids ascending inside a list, src/rii.h:356-358)
// ---------------------------------------------------------------------------------------------------

constexpr int kScanThreads = 1024;       // 16 waves: 4 per SIMD, 1 workgroup per CU (LDS-bound)
constexpr int kMaxLutLdsBytes = 144 * 1024;

struct ScanParams {
    const uint8_t *codes;        // [n_codes][M]
    int64_t n_codes;
    int M, Ks;
    const float *lut;            // tile-interleaved, see above
    int B;                       // number of real queries (tiles are padded)
    int QT;
    int chunks;                  // grid.x
    int64_t chunk_len;           // codes per chunk
    unsigned long long *best;    // [B] packed (orderable dist bits << 32 | local index), pre-set to ~0
    unsigned long long *keys;    // optional [Bc][n_codes] packed keys (general top-k path), else nullptr
    int b0, bc;                  // query range written to `keys`
    const int32_t *perm = nullptr;   // top-1 only: codes are in the LDS-friendly scan order, perm[pos] = id
};

// orderable mapping of an fp32 to u32 (monotone for all finite values incl. negatives)
__host__ __device__ inline uint32_t f32_orderable(uint32_t bits)
{
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}
__host__ __device__ inline uint32_t f32_unorderable(uint32_t u)
{
    return (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
}

// ---- kernel launchers (kernels.hip) ---------------------------------------------------------------
hipError_t launch_lut_build(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ks, int Ds,
                            int arch, int QT, float *d_lut, hipStream_t st);
hipError_t launch_lut_build_mfma(const float *d_queries, int64_t B, const float *d_codewords,
                                 const float *d_cnorm, int M, int Ks, int Ds, int QT, float *d_lut,
                                 hipStream_t st);
hipError_t launch_codeword_norms(const float *d_codewords, int M, int Ks, int Ds, float *d_cnorm, hipStream_t st);
hipError_t launch_lut_untile(const float *d_lut, int64_t B, int M, int Ks, int QT, float *d_out, hipStream_t st);

hipError_t launch_scan(const ScanParams &p, hipStream_t st);
hipError_t launch_finalize_top1(const unsigned long long *d_best, int64_t B, const int64_t *d_remap,
                                int64_t *d_out_ids, float *d_out_dists, int topk, hipStream_t st);
hipError_t launch_gather_sorted_topk(const unsigned long long *d_sorted, int64_t bc, int64_t n_codes, int topk,
                                     const int64_t *d_remap, int64_t *d_out_ids, float *d_out_dists,
                                     hipStream_t st);
hipError_t launch_gather_codes(const uint8_t *d_codes, int M, const int64_t *d_ids, int64_t S, uint8_t *d_out,
                               hipStream_t st);
hipError_t launch_gather_codes_i32(const uint8_t *d_codes, int M, const int32_t *d_ids, int64_t S, uint8_t *d_out, hipStream_t st);

// IVF
struct IvfParams {
    const uint8_t *codes; int64_t N; int M, Ks;
    const float *lut; int QT;
    const uint8_t *centers; int nlist;
    const int64_t *pl_off;        // [nlist+1] offsets into ids buffers
    const int32_t *pl_ids;        // ids actually traversed (filtered copy when S>0)
    const uint8_t *lcodes = nullptr;  // round 4, ivf_fused_kernel only: the codes in the order of pl_ids (row pp = code of posting pp); NULL = gather by id
    const int32_t *list_len;      // [nlist] lengths actually traversed (filtered when S>0)
    int64_t B; int b0;            // queries [b0, b0+B) of the batch are processed by this launch group
    int topk; int64_t L; int64_t w;
    float *coarse_dist;           // [B][nlist]
    int32_t *coarse_id;           // [B][nlist]
    int32_t *cum;                 // [B][nlist+1]
    int32_t *ncand;               // [B]
    int32_t *nvis;                // [B]
    int32_t *cand_id;             // [B][cand_stride]
    float *cand_dist;             // [B][cand_stride]
    int64_t cand_stride;
    int64_t *out_ids; float *out_dists; int64_t *out_counts;   // rows b0.. of the caller's outputs
    const float *queries;         // non-null: ivf_fused_kernel builds the table itself from (queries, codewords)
    const float *codewords; int Ds; int arch;
    int sel_cap;                  // ivf_fused_kernel: capacity of its list-selection array (ivf_fused_sel_cap)
    int32_t *flag_list; int *nflag;  // compact list of flagged queries + its length (filled by ivf_fused_kernel)
    int *nflag_next = nullptr;       // the counter of the NEXT launch group: zeroed by ivf_fused_kernel
    int kcap = 0;                    // ivf_fused_kernel, selection in LDS: keys of the final sort (set by launch_ivf_fused)
    int32_t *flag;                // [B] 1 = needs the exact std::partial_sort emulation path (nullptr = all do)
    int force_flag = 0;           // debug/tests: ivf_fused_kernel flags every query (option "ivf_force_exact")
    // host_spin: the outputs (and `flag`) are coherent HOST memory; ivf_fused_kernel stores host_seq into host_flag[b0 + b] once
    // query b's rows (or its fallback flag) are written, behind a system-scope release
    unsigned int *host_flag = nullptr; unsigned int host_seq = 0;
    // round 4: ivf_fused_kernel redoes a flagged query itself (ivf_exact_big_query): one global scratch slice per query of the launch
    // group (ivf_exact_big_scratch() bytes each) and the LDS heap capacity; NULL = hand over to the flag-gated exact kernels
    unsigned char *inl_scratch = nullptr; size_t inl_per_q = 0; int inl_hcap = 0;
    int q_host_off = 0;           // != 0: `queries` is coherent HOST memory (Ds = 4, Ks = 256): fetched once per block into LDS (the launcher
                                  // turns the flag into the byte offset of that staging area)
    // round 6, ivf_rot_kernel only: the centres and the posting-order codes once more, in tiles of 64 rows with row r rotated by
    // (r mod 64) mod min(M, 32) bytes (stored byte j = code[(j - rot) mod M]); every list starts on a tile boundary of rlcodes
    // (rl_toff[i] = its first tile), the last tile of a list is zero-filled
    const uint8_t *rcent = nullptr; const uint8_t *rlcodes = nullptr; const int32_t *rl_toff = nullptr;
};
hipError_t launch_ivf_coarse(const IvfParams &p, hipStream_t st);
bool ivf_fused_supported(int M, int Ks, int nlist, int64_t w, int topk);
int ivf_fused_sel_cap(int nlist, int64_t w);
hipError_t launch_ivf_fused(const IvfParams &p, hipStream_t st);
// round 5: four queries per block (tables interleaved [m][ks][query]: one ds_read_b128 scores a centre for four queries); top-1,
// Ds = 4, Ks = 256, M = 16 / 32, nlist <= 1024, w <= 32; same IvfParams, same flag protocol, no host_flag / q_host_off
bool ivf_quad_supported(int M, int Ks, int Ds, int nlist, int64_t w, int topk);
hipError_t launch_ivf_quad(const IvfParams &p, hipStream_t st);
// round 6: conflict-free table gather (table [ks][64 columns], lanes skewed in time: ivf_rot_kernel in kernels.hip); top-1, Ks = 256,
// M = 32 / 64, Ds = 2 / 4, nlist <= 1024, w <= 32, unfiltered lists (needs rcent / rlcodes / rl_toff); same flag protocol
bool ivf_rot_supported(int M, int Ks, int Ds, int nlist, int64_t w, int topk);
bool ivf_rot_fits(int64_t L, int64_t w);
hipError_t launch_ivf_rot(const IvfParams &p, hipStream_t st);
// the rotated tile copies: rows [n_rows][M] -> tiles; lists: list i's rows pl_off[i] .. of `lcodes` -> tiles rl_toff[i] ..
hipError_t launch_rot_rows(const uint8_t *d_src, int64_t n_rows, int M, uint8_t *d_dst, int64_t n_tiles, hipStream_t st);
hipError_t launch_rot_lists(const uint8_t *d_lcodes, const int64_t *d_pl_off, const int32_t *d_rl_toff, int nlist, int M, uint8_t *d_dst,
                            int64_t n_tiles, hipStream_t st);
hipError_t launch_ivf_plan(const IvfParams &p, hipStream_t st);
hipError_t launch_ivf_scan(const IvfParams &p, hipStream_t st);
hipError_t launch_ivf_select(const IvfParams &p, hipStream_t st);
bool ivf_exact_lds_supported(int M, int Ks, int nlist, int64_t L);
hipError_t launch_ivf_exact_lds(const IvfParams &p, hipStream_t st);
// shapes past the LDS kernel's limits: sequences in global scratch, heaps in LDS (any nlist, any L; w, topk <= 1024)
bool ivf_exact_big_supported(int M, int Ks, int64_t w, int topk);
int ivf_exact_big_heap_cap(int64_t w, int topk);             // entries of the LDS heap of ivf_exact_big_query
size_t ivf_exact_big_scratch(int nlist, int64_t L);          // bytes per block of the grid
hipError_t launch_ivf_exact_big(const IvfParams &p, void *d_scratch, int grid, hipStream_t st);
hipError_t launch_bitmap_set(const int64_t *d_tids, int64_t S, uint32_t *d_bitmap, hipStream_t st);
hipError_t launch_filter_lists(const int64_t *d_pl_off, const int32_t *d_pl_ids, int nlist,
                               const uint32_t *d_bitmap, int32_t *d_fids, int32_t *d_flen, hipStream_t st);

// coarse assignment / PQk-means
hipError_t launch_symtab(const float *d_codewords, int M, int Ks, int Ds, int arch, float *d_symtab,
                         hipStream_t st);
hipError_t launch_assign(const uint8_t *d_codes, int64_t num, int M, int Ks, const float *d_symtab,
                         const uint8_t *d_centers, int nlist, int32_t *d_assign, hipStream_t st);
hipError_t launch_pqk_hist(const uint8_t *d_data, const int32_t *d_assign, int64_t n, int M, int Ks,
                           int32_t *d_hist, int32_t *d_cnt, hipStream_t st);
hipError_t launch_pqk_vote(const int32_t *d_hist, const int32_t *d_cnt, const float *d_symtab, int K, int M,
                           int Ks, uint8_t *d_centers, hipStream_t st);

// segmented sort of packed keys (sortsel.hip, rocPRIM): sorts each of `segs` rows of length `len`
hipError_t segmented_sort_keys(unsigned long long *d_keys_in, unsigned long long *d_keys_out, int64_t segs,
                               int64_t len, void **d_temp, size_t *temp_bytes, hipStream_t st);

int lut_tile_for(int M, int Ks);

// fast scan (fastscan.hip): 8-bit filter + exact re-rank, top-1
bool fastscan_supported(int M, int Ks);
int fastscan_rows(int M, int Ks);
hipError_t launch_lut_quantize(const float *d_lut, int64_t B, int M, int Ks, int QT, uint8_t *d_qc, uint8_t *d_qlut,
                               int32_t *d_slack, int mx, hipStream_t st, int levels = 63);     // levels: 63, or 127 / 255 for fscan_mx_*
hipError_t launch_lut_build_quant(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ks, int Ds,
                                  int arch, float *d_lut, uint8_t *d_qc, uint8_t *d_qlut, int32_t *d_slack,
                                  unsigned int *d_cand_cnt, uint32_t *d_gthr, int mx, hipStream_t st, int levels = 63);
bool fs_rot_supported(int M, int Ks, int mx);
// round 4: the top-1 re-rank folded into the filter scan's tail (fscan_mx_kernel / fscan_mx_dual_kernel, M = 16 / 32, Ks = 256,
// Ds = 4 / 2: the shapes of rerank_top1_direct_kernel).  The chunk-blocks of a tile publish their candidates write-through and
// count themselves on tile_done[blockIdx.y]; the LAST one re-ranks the tile's queries from the codebook and writes the rows --
// into device memory, or straight into coherent host memory with host_flag[blockIdx.y] = seq raised behind them.
struct FsTail {
    const float *queries = nullptr;       // non-null: enabled
    const float *codewords = nullptr;
    const uint8_t *codes = nullptr;       // what a candidate's position indexes: the plain codes (id order)
    const int64_t *remap = nullptr;       // subset search: position -> id (also the code index when `indirect`)
    int indirect = 0, Ds = 4, topk = 1;
    int64_t *out_ids = nullptr;
    float *out_dists = nullptr;
    unsigned int *tile_done = nullptr;    // [grid.y] arrival counters: zero between launches (the last block puts the zero back)
    unsigned int *host_flag = nullptr;    // [grid.y] (coherent host memory) or NULL
    unsigned int seq = 0;
};
bool fscan_tail_supported(int M, int Ks, int Ds, int mx);
int fscan_tail_flags(int M, int Ks, int mx, int dual, int64_t B);       // number of tile_done counters / host flags a launch uses
hipError_t launch_fscan(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const uint8_t *d_qlut,
                        const int32_t *d_slack, int B, int chunks, int64_t chunk_len, unsigned long long *d_cand,
                        unsigned int *d_cand_count, int cap, int mode, uint16_t *d_segmin, const uint32_t *d_thr16,
                        uint32_t *d_gthr, int sample_stride, int mx, hipStream_t st, int quarter = 0, int dual = 0, int levels = 63,
                        const FsTail *tail = nullptr, int pipe = 1);
// queries per block of the filter scan: 32 when the M = 16 shape runs two tiles per block (option scan_dual), else fastscan_rows()
int fscan_queries_per_block(int M, int Ks, int mx, int dual);
int fastscan_max_sum(int M, int levels = 0);      // largest quantised sum: M x levels (0 = the default 63)
// round 3: the tables of the matrix-core filter in ONE launch (quarter tables, d_lut_or_null = also the exact fp32 table), and
// the top-1 re-rank that needs no table in global memory
bool qlut_fused_supported(int M, int Ks, int Ds, int mx);
size_t qlut_fused_bytes(int64_t B, int M);
hipError_t launch_qlut_fused(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ds, float *d_lut_or_null,
                             uint32_t *d_qlut4, int32_t *d_slack, unsigned int *d_cand_cnt, uint32_t *d_gthr, int levels, hipStream_t st);
bool rerank_direct_supported(int M, int Ks, int Ds);
hipError_t launch_rerank_top1_direct(const uint8_t *d_codes, int64_t n_codes, int M, int Ds, const float *d_queries,
                                     const float *d_codewords, const int32_t *d_slack, const unsigned long long *d_cand,
                                     const unsigned int *d_cand_count, int cap, const int64_t *d_remap, int64_t B,
                                     int64_t *d_out_ids, float *d_out_dists, int topk, int indirect, hipStream_t st);
// mx != 0: the rotated shapes run fscan_mx_kernel (byte sums on the matrix cores) over its own order of the formatted lookups
int fscan_segments_per_chunk(int M, int Ks, int mx);
int fscan_mx_subspace(int M, int lane, int t);       // subspace whose table row lane `lane` of a wave fetches as its lookup t
int64_t fcodes_bytes(int64_t n, int M, int mx);      // size of the formatted copy of n codes
// tables of the rotated shapes built by tile (fastscan.hip): exact fp32 [b][M*Ks] + rotated byte rows + slack in two launches
bool lut_tile_supported(int M, int Ks, int Ds, int mx);
hipError_t launch_lut_tile_build_quant(const float *d_queries, int64_t B, const float *d_codewords, int M, int Ds, float *d_lut,
                                       float *d_lohi, uint8_t *d_qlut, int32_t *d_slack, unsigned int *d_cand_cnt,
                                       uint32_t *d_gthr, hipStream_t st);
// conflict-free rotated table layout + formatted code copy (see fastscan.hip): launch_fscan then takes the formatted codes
bool fs_rot_supported(int M, int Ks, int mx);      // mx: the engine's scan_mx option (M = 64 has a rotated form only there)
hipError_t launch_fcodes_format(const uint8_t *d_codes, const int64_t *d_ids, int64_t n0, int64_t n1, int M, int Ks,
                                uint16_t *d_out, int mx, hipStream_t st);
int rerank_topk_max_k();
hipError_t launch_kth_threshold(const uint16_t *d_segmin, int64_t G, int64_t B, int k, int maxv, const int32_t *d_slack,
                                uint32_t *d_thr16, hipStream_t st);
hipError_t launch_rerank_topk(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const float *d_lut, int QT,
                              const unsigned long long *d_cand, const unsigned int *d_cand_count, int cap,
                              const int64_t *d_remap, const int32_t *d_perm, int64_t B, int64_t *d_out_ids,
                              float *d_out_dists, int topk, int32_t *d_flag_list, int *d_nflag, int indirect, hipStream_t st);
hipError_t launch_rerank_top1(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const float *d_lut, int QT,
                              const int32_t *d_slack, const unsigned long long *d_cand,
                              const unsigned int *d_cand_count, int cap, const int64_t *d_remap,
                              const int32_t *d_perm, int64_t B, int64_t *d_out_ids, float *d_out_dists, int topk,
                              int indirect, hipStream_t st, unsigned int *d_peak = nullptr);   // QT in {4,2,1,0}; 0 = table does not fit LDS (unsupported shape)

// tieorder.hip: the reference's std::partial_sort order for the queries whose k+1 smallest distances tie exactly
bool linear_tie_supported(int M, int Ks);
bool linear_tie_heap_in_lds(int M, int Ks, int topk);
hipError_t launch_linear_tie(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, int QT, int64_t b0,
                             const int32_t *d_flag_list, const int *d_nflag, const int64_t *d_remap, int64_t *d_out_ids,
                             float *d_out_dists, int topk, int grid, unsigned long long *d_heap, int indirect, int first, hipStream_t st);
bool linear_tie_chunked_supported(int M, int Ks, int topk);
size_t linear_tie_chunked_scratch(int64_t n, int fq);
hipError_t launch_linear_tie_chunked(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, int QT, int64_t b0,
                                     const int32_t *d_flag_list, const int *d_nflag, const int64_t *d_remap, int64_t *d_out_ids,
                                     float *d_out_dists, int topk, int fq, void *d_scratch, int indirect, hipStream_t st,
                                     const float *d_queries = nullptr, const float *d_codewords = nullptr, int Ds = 0, int arch = 0,
                                     int *d_nflag_next = nullptr);
hipError_t launch_sorted_tie_flag(const unsigned long long *d_sorted, int64_t bc, int64_t n_codes, int topk,
                                  int32_t *d_flag_list, int *d_nflag, hipStream_t st);

// ivfshard.hip: inverted-index search over a database-sharded index (global stop rule from all-gathered list lengths).  Any L
// (round 5): rows = k + 1 best owned candidates per query while rows <= ivf_shard_max_select_rows(), or rows >= L = EVERY owned
// candidate at the slot of its traversal position (exact-tie replay; the collect-all route of large k)
bool ivf_shard_supported(int M, int Ks, int nlist, int64_t L, int64_t w, int rows);
int ivf_shard_max_select_rows(int M, int Ks, int nlist, int64_t L, int64_t w);
size_t ivf_shard_scratch_per_query(int M, int Ks, int nlist, int64_t L, int64_t w);   // global scratch per query of a launch (0: everything fits LDS)
// the exchange record of rii_query_ivf_dbsharded_dev written by the shard kernel itself ([n] int64 positions | [n] int64 global ids | [n] f32)
struct ShardPack { int64_t *rec_pos = nullptr, *rec_id = nullptr; float *rec_d = nullptr; int64_t id_offset = 0; int32_t *zero2 = nullptr; };
hipError_t launch_ivf_shard(const uint8_t *d_codes, int M, int Ks, const float *d_lut, const uint8_t *d_centers, int nlist,
                            const int64_t *d_pl_off, const int32_t *d_pl_ids, const int32_t *d_list_len, const int32_t *d_glen,
                            int G, int rank, int64_t B, int topk, int64_t L, int64_t w, int rows, int64_t *d_out_ids, float *d_out_dists,
                            int32_t *d_out_pos, int32_t *d_out_nloc, int64_t *d_out_counts, void *d_scratch, hipStream_t st,
                            const float *d_queries = nullptr, const float *d_codewords = nullptr, int Ds = 0, int arch = 0,
                            const uint8_t *d_lcodes = nullptr, int debug = 0,       // d_lcodes: the codes in posting order of d_pl_ids (unfiltered lists), or NULL
                            const unsigned long long *d_picks = nullptr, const int32_t *d_pick_ok = nullptr,     // the pre-pass's output (with d_lut), or NULL
                            const ShardPack *pack = nullptr);
// round 6: the batch's coarse phase in front of that launch -- four queries per block, tables interleaved [m][ks][query]
// (shard_coarse_quad_kernel): writes the queries' plain tables to d_lut ([B][M * 256]), the w + 1 smallest (distance, list) keys per
// query to d_picks ([B][kShardPickStride]) and whether they are conclusive to d_pick_ok ([B])
constexpr int kShardPickStride = 8;
bool shard_coarse_supported(int M, int Ks, int Ds, int nlist, int64_t L, int64_t w, int rows);
hipError_t launch_shard_coarse(const float *d_queries, const float *d_codewords, const uint8_t *d_centers, int M, int Ds, int arch, int nlist,
                               int64_t w, int64_t B, float *d_lut, unsigned long long *d_picks, int32_t *d_pick_ok, hipStream_t st, int debug = 0);
// true: that launch builds the queries' tables itself when handed (d_queries, d_codewords) -- d_lut may be NULL then
bool ivf_shard_builds_tables(int M, int Ks, int nlist, int64_t L, int64_t w, int rows);
size_t shard_replay_scratch(int64_t nf, int rows);    // bytes of d_scratch launch_shard_replay needs (0: the sequences fit LDS)
hipError_t launch_shard_replay(const void *d_gathered, int G, int64_t nf, int rows, int topk, int64_t *d_out_ids,
                               float *d_out_dists, void *d_scratch, hipStream_t st);

// merge.hip: database sharding, k-way merge of the gathered per-shard top-k rows under (dist, id)
int merge_topk_max_keys();
size_t merge_record_bytes(int64_t B, int k, int payload);
// id_offsets: host array of G per-rank offsets added to the (non-padding) keys, or NULL; d_out_tie [B] / d_out_any [1]: optional
// tie flags over the first tie_cols merged distances (d_out_any must be zeroed by the caller)
// round 6: the sharded inverted index's top-1 batch (k = 2 rows per query and rank, payload records with 16-byte headers): merge under
// (distance, position) + the finishing step in one launch, a thread per query
hipError_t launch_ivf_merge_top1(const void *d_gathered, int G, int64_t B, int hdr, const int64_t *d_cnt, int64_t *d_out_ids, float *d_out_d,
                                 int64_t *d_out_cnt, int32_t *d_out_tie, int32_t *d_out_any, hipStream_t st);
hipError_t launch_merge_topk(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, int64_t *d_out_ids,
                             float *d_out_dists, int64_t *d_out_payload, hipStream_t st, const int64_t *id_offsets = nullptr,
                             int tie_cols = 0, int32_t *d_out_tie = nullptr, int32_t *d_out_any = nullptr, int hdr = 0,
                             void *d_scratch = nullptr);
// G * k above merge_topk_max_keys(): the keys are sorted in d_scratch (merge_topk_scratch() bytes) instead of LDS -- any G, any k
size_t merge_topk_scratch(int G, int64_t B, int k);
// hdr = kRecHeader: every rank's record is preceded by {int64 id offset of its shard, int32 status, pad} (merge.hip); a non-zero
// status on any rank poisons every row of the batch on every rank (id -2, distance NaN; bit 1 of *d_out_any)
constexpr int kRecHeader = 16;

// smalltopk.hip: a small batch over a small index, one launch
// d_lut == NULL: every block builds its exact table from d_queries and the codebook itself (no table launch).
// small_topk_slices() > 1: the codes of a query are scored by that many blocks; d_keys = small_topk_scratch() bytes, d_done = B
// counters that are zero between launches (zeroed once by the caller, put back by the kernel).
bool small_topk_supported(int M, int Ks, int Ds, int64_t n, int topk);
int small_topk_slices(int64_t n, int64_t B);
size_t small_topk_scratch(int64_t n, int64_t B);
hipError_t launch_small_topk(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, const float *d_queries,
                             const float *d_codewords, int Ds, int arch, int64_t B, int topk, const int64_t *d_remap,
                             unsigned long long *d_keys, unsigned int *d_done, int64_t *d_out_ids, float *d_out_dists, hipStream_t st,
                             unsigned int *host_flag = nullptr, unsigned int seq = 0);   // host_flag: outputs in coherent host memory

// smalltopk.hip: a few queries per HOST call on a large index, one launch (slices of the codes on all CUs, last block merges);
// d_out_tie[b] = 1: two of the k+1 smallest distances tie -- the caller reruns the call on the general path.  d_cand =
// slice_topk_scratch() bytes, d_done = B counters that are zero between launches.
bool slice_topk_supported(int M, int Ks, int Ds, int64_t n, int64_t B, int topk);
size_t slice_topk_scratch(int64_t n, int64_t B, int topk);
hipError_t launch_slice_topk(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_queries, const float *d_codewords, int Ds, int arch,
                             int64_t B, int topk, const int64_t *d_remap, unsigned long long *d_cand, unsigned int *d_done,
                             int64_t *d_out_ids, float *d_out_dists, int32_t *d_out_tie, hipStream_t st, unsigned int *host_flag, unsigned int seq,
                             int32_t *d_flag_list = nullptr, int *d_nflag = nullptr);   // d_flag_list / d_nflag: tied queries appended (device-side fallback)

// widetab.hip: shapes whose one-query table does not fit LDS (lut_tile_for() == 0): tables stay in global memory
hipError_t launch_scan_wide(const uint8_t *d_codes, int64_t n_codes, int M, int Ks, const float *d_lut, const int64_t *d_remap,
                            int b0, int bc, unsigned long long *d_keys, hipStream_t st);
hipError_t launch_tie_rows(unsigned long long *d_keys, int64_t n_codes, int64_t b0, int64_t bc, const int32_t *d_flag_list, const int *d_nflag,
                           int topk, const int64_t *d_remap, int64_t *d_out_ids, float *d_out_dists, hipStream_t st);
hipError_t launch_assign_wide(const uint8_t *d_codes, int64_t num, int M, int Ks, const float *d_symtab, const uint8_t *d_centers, int nlist,
                              int32_t *d_assign, hipStream_t st);

// scanorder.hip: LDS-friendly scan order for the filter stage (perm[pos] = code id, codes gathered in that order)
bool scan_order_supported(int M, int Ks);
hipError_t launch_scan_order(const uint8_t *d_codes, int64_t N, int M, int Ks, int rows, int64_t win0, int32_t *d_perm,
                             uint8_t *d_out_codes, hipStream_t st);

// database-sharded linear search, exact ties (tieorder.hip): candidate lists per shard + replay over the gathered lists
hipError_t launch_linear_tie_emit(const uint8_t *d_codes, int64_t n, int M, int Ks, const float *d_lut, int QT, int64_t b0,
                                  const int32_t *d_flag_list, const int *d_nflag, const int64_t *d_remap, int topk, int fq,
                                  void *d_scratch, int indirect, const float *d_ext_bound, int64_t id_offset, int cap,
                                  int64_t *d_out_ids, float *d_out_dists, int32_t *d_out_count, hipStream_t st,
                                  const unsigned long long *d_keyrow = nullptr);
bool linear_tie_chunked_topk_ok(int topk);
size_t linear_tie_record_bytes(int64_t nf, int cap);
hipError_t launch_linear_shard_replay(const void *d_gathered, int G, int64_t nf, int cap, int topk, int64_t *d_out_ids,
                                      float *d_out_dists, hipStream_t st);
// comm.hip (round 4): the exchange step behind the C ABI -- RCCL bound at run time (dlopen), the packed-record all-gather, and the
// kernels around it.  The const char* results are error messages (NULL = success).
const char *rccl_load_error();
const char *comm_unique_id(void *out128);
const char *comm_create(const void *id128, int rank, int G, void **out_comm);
void comm_destroy(void *comm);
const char *comm_all_gather(void *comm, const void *d_send, void *d_recv, size_t bytes, hipStream_t st);
int64_t qshard_begin(int64_t B, int G, int r);                                 // first row of rank r's slice of a batch of B
size_t qshard_record_bytes(int64_t B, int G, int k, int counts);
hipError_t launch_qshard_unpack(const void *d_gathered, int64_t B, int G, int k, int counts, int64_t *d_out_ids, float *d_out_dists,
                                int64_t *d_out_counts, hipStream_t st);
hipError_t launch_merge_top1(const void *d_gathered, int G, int64_t B, const int64_t *id_offsets, int64_t *d_out_ids, float *d_out_dists,
                             hipStream_t st, int hdr = 0);
hipError_t launch_tie_prepare(const void *d_gathered, int rank, int64_t B, int rows, int topk, const int32_t *d_fsel, int nf,
                              const float *d_queries, int D, float *d_qf, float *d_bound, hipStream_t st, int hdr = 0);
hipError_t launch_tie_scatter(const void *d_gg, int G, int nf, int cap, int topk, const int32_t *d_fsel, const int64_t *d_r_ids,
                              const float *d_r_d, int64_t *d_out_ids, float *d_out_dists, int32_t *d_overflow, hipStream_t st);
hipError_t launch_copy_cols(const int64_t *d_in_i, const float *d_in_d, int64_t B, int in_stride, int out_stride, int ncols, int64_t *d_out_i,
                            float *d_out_d, hipStream_t st);
hipError_t launch_fill_pad(int64_t *d_ids, float *d_d, int64_t n, hipStream_t st);
hipError_t launch_ivf_pack(const int64_t *d_ids, const int32_t *d_pos, const float *d_d, int64_t n, int64_t id_offset, void *d_rec, hipStream_t st);
hipError_t launch_ivf_finish(const int64_t *d_mi, const float *d_md, const int64_t *d_cnt, int64_t B, int k1, int topk, int64_t *d_out_ids,
                             float *d_out_d, int64_t *d_out_cnt, int32_t *d_tie, int32_t *d_any, hipStream_t st);
hipError_t launch_gather_rows(const float *d_src, const int32_t *d_fsel, int nf, int D, float *d_dst, hipStream_t st);
hipError_t launch_scatter_rows(const int32_t *d_fsel, int nf, int k, const int64_t *d_r_i, const float *d_r_d, int64_t *d_out_i, float *d_out_d,
                               hipStream_t st);
}  // namespace riiamd
