/*
 * rii_amd.h -- C ABI of the MI355X-native IVFPQ query engine (drop-in for the hot path of matsui528/rii).
 *
 * One opaque engine == one `rii::RiiCpp` (reference: src/rii.h:40-83) living on one GPU.  Every entry
 * point below names the reference interface it replaces (paths relative to the reference repo).  Plain
 * pointers and sizes only; the caller allocates every output.  All functions return RII_OK (0) or a
 * negative error code; rii_last_error() gives the message (thread-local).  Where the reference aborts the
 * process (live `assert`, bare `throw;` -- src/rii.h:110-111,166-170,202,219-220,252-253) this library
 * returns RII_ERR_INVALID / RII_ERR_STATE instead.
 *
 * Host-pointer calls copy their inputs during the call exactly like the reference (src/rii.h:93-100,
 * 177-182); nothing outlives a call.  The *_dev variants take device pointers (HBM-resident batches) and
 * an optional hipStream_t (as void*, NULL = the engine's stream) and are asynchronous on that stream.
 *
 * NEW relative to the reference: every query entry point takes a batch of B queries (the reference is one
 * query per call: src/main.cpp:17-27).  Row b of the outputs is exactly what the reference returns for
 * query b alone.
 */
#ifndef RII_AMD_H
#define RII_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rii_engine rii_engine;

enum {
    RII_OK = 0,
    RII_ERR_INVALID = -1,      /* violated precondition (the reference's assert()s) */
    RII_ERR_STATE = -2,        /* e.g. add_codes(update_flag=1) before reconfigure (src/rii.h:166-170) */
    RII_ERR_HIP = -3,          /* a HIP runtime call failed / no GPU: the product path never falls back */
    RII_ERR_UNSUPPORTED = -4
};

/* Which compile-time variant of fvec_L2sqr (src/distance.h:113,172,219) and of the auto-vectorised
 * PQk-means table loop (src/pqkmeans.cpp:164-173) the reference build being replaced would have used
 * under its `-march=native` (setup.py:96).  Only matters for Ds >= 8 (LUT) / Ds >= 4 (coarse tables). */
enum { RII_SIMD_SSE = 0, RII_SIMD_AVX = 1, RII_SIMD_AVX512 = 2 };

/* How the per-query distance table (LUT) is built. */
enum {
    RII_LUT_EXACT = 0,   /* VALU, the reference's exact lane/FMA order: bit-identical distances (default) */
    RII_LUT_MFMA = 1     /* v_mfma_f32_16x16x4_f32 on |q|^2 - 2 q.c + |c|^2: within 1e-4 relative */
};

const char *rii_last_error(void);
const char *rii_version(void);
int rii_device_count(void);

/* RiiCpp(codewords[M,Ks,Ds], verbose) -- src/main.cpp:14, src/rii.h:86-106.  Copies the codewords.
 * device: HIP device ordinal.  simd_arch: see above. */
int rii_create(const float *codewords, int M, int Ks, int Ds, int verbose, int simd_arch, int device,
               rii_engine **out);
void rii_destroy(rii_engine *e);

/* RiiCpp::AddCodes(codes[n,M], update_flag) -- src/main.cpp:16, src/rii.h:158-193. */
int rii_add_codes(rii_engine *e, const uint8_t *codes, int64_t n, int update_flag);
/* RiiCpp::Reconfigure(nlist, iter) -- src/main.cpp:15, src/rii.h:108-156 (sampling, PQk-means fit on the
 * GPU, posting-list rebuild).  Requires 0 < nlist <= N. */
int rii_reconfigure(rii_engine *e, int nlist, int iter);
/* RiiCpp::Clear() -- src/main.cpp:28, src/rii.h:328-333. */
int rii_clear(rii_engine *e);

/* State import: what the reference does through py::pickle's set-state (src/main.cpp:39-52).
 * rii_set_coarse_centers replaces the centres and rebuilds all posting lists by coarse assignment
 * (src/rii.h:150-155); rii_set_state installs centres, codes and lists verbatim; rii_set_posting_lists (round 5) installs centres
 * and lists verbatim over the codes already added (ids ascending inside a list is the caller's contract, src/rii.h:356-358; the
 * ids are range-checked): lists built elsewhere -- a cached index as in examples/benchmark/run_sift1b.py:73-99, another rank --
 * without the N x nlist x M assignment pass. */
int rii_set_coarse_centers(rii_engine *e, const uint8_t *centers, int64_t nlist);
int rii_set_posting_lists(rii_engine *e, const uint8_t *centers, int64_t nlist, const int64_t *pl_off, const int32_t *pl_ids);
int rii_set_state(rii_engine *e, const uint8_t *centers, int64_t nlist, const uint8_t *codes, int64_t N,
                  const int64_t *pl_off, const int32_t *pl_ids);

/* Properties -- src/main.cpp:29-34.  Getters copy out. */
int64_t rii_get_N(const rii_engine *e);
int64_t rii_get_nlist(const rii_engine *e);
int rii_get_M(const rii_engine *e);
int rii_get_Ks(const rii_engine *e);
int rii_get_Ds(const rii_engine *e);
int rii_get_verbose(const rii_engine *e);
int rii_set_verbose(rii_engine *e, int verbose);
int rii_get_codewords(const rii_engine *e, float *out /* M*Ks*Ds */);
int rii_get_codes(const rii_engine *e, uint8_t *out /* N*M */);                 /* flattened_codes */
int rii_get_coarse_centers(const rii_engine *e, uint8_t *out /* nlist*M */);    /* coarse_centers */
int rii_get_posting_lists(const rii_engine *e, int64_t *off /* nlist+1 */, int32_t *ids /* N */);

/* RiiCpp::QueryLinear -- src/main.cpp:17-21, src/rii.h:195-242, for B queries.
 * queries: [B, D] row-major fp32 (D = M*Ds).  tids: S int64 ids in [0, N) shared by the batch, S == 0 means "all";
 * scored in the order given, duplicates included, like the reference (the inverted index treats them as a set, which
 * is the reference's answer for the sorted ids it documents).  out_ids [B,topk] int64, out_dists [B,topk] fp32, ascending distance.
 * Requires topk <= N and (S == 0 || topk <= S <= N). */
int rii_query_linear(rii_engine *e, const float *queries, int64_t B, int topk, const int64_t *tids, int64_t S,
                     int64_t *out_ids, float *out_dists);
/* RiiCpp::QueryIvf -- src/main.cpp:22-27, src/rii.h:244-326, for B queries.
 * out_counts[b] = topk, or 0 where the reference returns ({}, {}) (src/rii.h:324-325).
 * Requires topk <= L <= N and (S == 0 || topk <= S <= N). */
int rii_query_ivf(rii_engine *e, const float *queries, int64_t B, int topk, const int64_t *tids, int64_t S,
                  int64_t L, int64_t *out_ids, float *out_dists, int64_t *out_counts);

/* The same with every pointer in device memory (HBM-resident batch); asynchronous on `stream`. */
int rii_query_linear_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                         int64_t S, int64_t *d_out_ids, float *d_out_dists, void *stream);
int rii_query_ivf_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                      int64_t S, int64_t L, int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts,
                      void *stream);
/* (Memory: the first unfiltered inverted-index query after the lists change builds a posting-order copy of the codes -- option
 * "ivf_list_codes", another N * M bytes of HBM, one synchronisation of the stream; where that allocation does not fit the kernels
 * gather the candidates' rows by id instead, same results.) */

/* Queries resident in HBM, results delivered to the HOST (NEW): the batched form of what the reference's caller observes --
 * the rows are host-visible when the call returns (src/main.cpp:17-27 return Python lists) -- and what SURVEY 8d's metric times
 * ("one batched call incl. table build, scan, top-k, device->host of results").  d_queries / d_tids are device pointers, out_* HOST
 * pointers (any memory).  The kernels write the rows straight into the engine's pinned, coherent host block; where the last kernel
 * of the step raises sequence flags behind them (linear top-1 of the M = 16 / 32, Ks = 256 shapes: the re-rank is the tail of the
 * scan launch, one flag per tile) the call returns as soon as the host has seen the flags -- no D2H copy, no stream
 * synchronisation -- otherwise after one hipStreamSynchronize.  Synchronous; `stream` as for the *_dev calls. */
int rii_query_linear_dev_to_host(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S,
                                 int64_t *out_ids, float *out_dists, void *stream);
int rii_query_ivf_dev_to_host(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S,
                              int64_t L, int64_t *out_ids, float *out_dists, int64_t *out_counts, void *stream);

/* Database-sharded inverted index (NEW, not in the reference: it has no multi-device code).  Rank `rank` of G holds
 * a contiguous id range of the database; coarse centres are replicated (rii_set_coarse_centers), posting lists are local.
 * The reference's "candidates in list order, stop at exactly L" rule (src/rii.h:283-321) is global and sequential, so
 *   1. every rank reports the lengths of its lists after the batch's target-id filter (rii_ivf_list_lengths_dev:
 *      nlist int32, `d_tids` = the S target ids that fall into this rank's range, as LOCAL ids; S_global = size of the
 *      whole target set, 0 = no filter -- a rank may own none of the targets: S == 0 with S_global != 0),
 *   2. the lengths are all-gathered ([G][nlist] int32, `d_glen`, identical on every rank), and
 *   3. rii_query_ivf_shard_dev replays the traversal on the global lengths and scores only the candidates this rank owns.
 * Outputs per query, `rows` rows (topk+1 by default) ascending by (distance, traversal position): LOCAL ids (-1 = none), distances (+inf),
 * traversal positions (INT32_MAX); d_out_nloc[b] = valid rows; d_out_counts[b] = topk, or 0 where the reference returns
 * ({}, {}) (identical on every rank).  Merging the G records under (distance, position) gives the reference's answer for
 * top-1 and for top-k whenever the k+1 smallest distances are pairwise different.  S_global / N_global: sizes of the whole
 * target set / database (they fix `w`, src/rii.h:266-277).  Any nlist <= N (above 4096 lists the coarse order of a
 * query lives in global scratch).  Any L <= N (round 5; the reference's billion-scale run uses L = N / nlist = sqrt(N) ~ 31.6 k,
 * examples/benchmark/run_sift1b.py:105-106): up to 8192 candidate keys of a query are sorted in LDS at once; above that the rank's
 * own candidates pass through an LDS selection buffer with a running bound.  `rows` (0 = topk + 1): the rows a launch can SELECT are
 * bounded by rii_ivf_shard_max_select_rows() (8193 while L <= 8192, 6144 above); rows >= L always works and returns EVERY owned
 * candidate -- in (distance, position) order, or (L > 6144) at the slot of its traversal position (row j = position j, other ranks'
 * slots padded): the replay below rebuilds the sequence by position either way.  (The coarse order of a query:
 * with w <= 7 and the order in global scratch the w + 1 smallest (distance, list) keys come from a register / DPP selection; the
 * library's std::partial_sort (src/rii.h:279-280) is replayed move for move only where its internals can show -- two of those w + 1
 * distances exactly equal, or a walk that continues past list w.  rows = 2, i.e. top-1, never
 * sorts: two keys per thread in registers.  Unfiltered lists: the candidates' rows come from the engine's posting-order copy of the
 * codes -- option "ivf_list_codes" -- one coalesced run per visited list; with target ids, or where the copy does not fit, by id.) */
int rii_ivf_list_lengths_dev(rii_engine *e, const int64_t *d_tids, int64_t S, int64_t S_global, int32_t *d_out_len,
                             void *stream);
int rii_query_ivf_shard_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S,
                            int64_t S_global, int64_t L, int64_t N_global, const int32_t *d_glen, int G, int rank, int rows,
                            int64_t *d_out_ids, float *d_out_dists, int32_t *d_out_pos, int32_t *d_out_nloc,
                            int64_t *d_out_counts, void *stream);
/* Exact ties: `rows` = output rows per query of rii_query_ivf_shard_dev (0 = topk + 1).  For a query whose merged k+1 best
 * distances hold an exact tie, call it again with rows = L (every candidate the rank owns), all-gather the records
 * ([nf*rows] int64 positions, [nf*rows] int64 GLOBAL ids, [nf*rows] f32 distances, padded to 16 bytes) and let
 * rii_ivf_shard_replay_dev rebuild the candidate sequence by position and replay std::partial_sort (src/rii.h:312-313) on it:
 * the reference's order, tied distances included, identically on every rank.  Stateless; nf = number of such queries. */
int rii_ivf_shard_replay_dev(const void *d_gathered, int G, int64_t nf, int rows, int topk, int64_t *d_out_ids,
                             float *d_out_dists, void *stream);
/* rows > 8192 (round 5): the rebuilt sequences live in caller-provided device scratch of rii_ivf_shard_replay_scratch_bytes(nf, rows)
 * bytes (0 while they fit LDS), the heap of std::partial_sort in LDS while topk <= 1024, walked by one lane over global memory above.
 * A query without candidates (the reference's ({}, {})) gets ids -1 / distances +inf.  With rows = L for EVERY query of a batch this
 * is the whole answer (no merge, no tie flags): the collect-all route rii_query_ivf_dbsharded_dev takes for very large topk. */
int64_t rii_ivf_shard_replay_scratch_bytes(int64_t nf, int rows);
int rii_ivf_shard_replay_ex_dev(const void *d_gathered, int G, int64_t nf, int rows, int topk, int64_t *d_out_ids,
                                float *d_out_dists, void *d_scratch, int64_t scratch_bytes, void *stream);
/* Largest `rows` rii_query_ivf_shard_dev selects per query for this engine's shape and (L, N_global, S_global); callers that need
 * more rows per query ask for rows = L instead (every owned candidate).  The same on every rank. */
int rii_ivf_shard_max_select_rows(const rii_engine *e, int64_t L, int64_t N_global, int64_t S_global);

/* Database-sharded LINEAR search, exact ties (NEW; tieorder.hip).  The merge below orders bit-equal distances of different
 * shards by id; the reference's order is what std::partial_sort (src/rii.h:234-235) makes of all N distances in index order.
 * For a query whose merged k+1 best distances hold an exact tie every rank calls rii_linear_tie_emit_dev: in index order, the
 * codes of its shard (or of its share of the target ids) that can touch the reference's heap -- distance below `d_bound[f]`
 * (+inf = none: the k-th smallest distance of any earlier shard with >= k codes is a valid bound) and below the k-th smallest
 * of every earlier 8192-code chunk of the shard -- as rows of `cap` (global id = id_offset + local id, distance), plus the
 * row's true length in d_out_count (> cap: the row is truncated and must not be used).  The ranks' records
 * ([nf] int32 counts padded to 8 bytes, [nf*cap] int64 ids, [nf*cap] f32 distances, padded to rii_linear_tie_record_bytes())
 * are all-gathered in rank order = index order, and rii_linear_tie_replay_dev (stateless) replays the library's heap over
 * them: the reference's ids, distances and order, identically on every rank.  topk <= 1024, G * cap < 2^32. */
int rii_linear_tie_emit_dev(rii_engine *e, const float *d_queries, int64_t nf, int topk, const int64_t *d_tids, int64_t S,
                            const float *d_bound, int64_t id_offset, int cap, int64_t *d_out_ids, float *d_out_dists,
                            int32_t *d_out_count, void *stream);
int64_t rii_linear_tie_record_bytes(int64_t nf, int cap);
int rii_linear_tie_replay_dev(const void *d_gathered, int G, int64_t nf, int cap, int topk, int64_t *d_out_ids,
                              float *d_out_dists, void *stream);

/* Database sharding (NEW, not in the reference: it has no multi-device code).  Every rank sends one record per batch --
 * [B*k] int64 keys, (payload != 0: [B*k] int64 payload,) [B*k] f32 distances, padded to rii_merge_record_bytes() -- through
 * one all-gather; `d_gathered` holds the G records back to back.  Output: per query the k_out smallest of the G*k entries
 * under (distance asc, key asc), computed identically on every rank, with their payloads.  Linear search: key = GLOBAL id,
 * no payload.  Inverted index: key = traversal position, payload = global id.  G*k <= 8192.  Asynchronous on `stream`
 * (a hipStream_t, NULL = the default stream) of the current device; needs no engine. */
int64_t rii_merge_record_bytes(int64_t B, int k, int payload);
int rii_merge_topk_dev(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, int64_t *d_out_keys,
                       float *d_out_dists, int64_t *d_out_payload, void *stream);
/* The same with (i) `id_offsets` (HOST array of G int64, or NULL): added to the keys of rank g's record while merging, so a
 * rank can hand the engine's LOCAL ids to the all-gather untouched (keys < 0 or >= 2^62 are padding and stay as they are);
 * (ii) `d_out_tie` [B] int32 (or NULL): 1 where two of the first `tie_cols` merged distances are bit-equal and finite -- the
 * queries whose order the caller must replay (rii_linear_tie_* above) -- and `d_out_any` [1] int32 (or NULL; zeroed by the
 * caller): OR of those flags, so that ONE 4-byte read tells whether any replay is needed.  G <= 64 with id_offsets. */
int rii_merge_topk_ex_dev(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, const int64_t *id_offsets,
                          int64_t *d_out_keys, float *d_out_dists, int64_t *d_out_payload, int tie_cols, int32_t *d_out_tie,
                          int32_t *d_out_any, void *stream);

/* The same merge over records that carry a 16-byte header {int64 id offset of the rank's shard, int32 status, int32 pad} in front of
 * their rows (round 5: the form the sharded entry points below exchange; rii_merge_hdr_record_bytes() = rii_merge_record_bytes() + 16):
 * any G, any k -- more than 8192 rows per query are sorted in `d_scratch` (rii_merge_hdr_scratch_bytes(), 0 below that) --, the
 * offsets are read from the headers and added to the non-padding keys, a non-zero status in ANY header poisons every row (keys /
 * payloads -2, distances NaN; bit 1 of *d_out_any).  Stateless. */
int64_t rii_merge_hdr_record_bytes(int64_t B, int k, int payload);
int64_t rii_merge_hdr_scratch_bytes(int G, int64_t B, int k);
int rii_merge_topk_hdr_dev(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, int64_t *d_out_keys, float *d_out_dists,
                           int64_t *d_out_payload, int tie_cols, int32_t *d_out_tie, int32_t *d_out_any, void *d_scratch,
                           int64_t scratch_bytes, void *stream);

/* Round 6: the merge AND the finishing step of the database-sharded inverted index's TOP-1 batch in one launch (a thread per query;
 * what rii_query_ivf_dbsharded_dev runs behind its all-gather when topk == 1).  Records as rii_merge_topk_hdr_dev's with payload
 * (rii_merge_hdr_record_bytes(B, 2, 1) per rank: keys = traversal positions, payload = global ids, k = 2 rows per query -- the two
 * best candidates each rank owns, padding rows: position INT32_MAX, distance +inf).  d_counts [B]: the global walk's verdict (> 0 =
 * found, the same on every rank: rii_query_ivf_shard_dev's counts).  Out: the first minimum under (distance, position), or -1 / +inf /
 * count 0 where the reference returns ({}, {}); a non-zero status in any header: ids -2, distances NaN, counts -1, bit 1 of *d_out_any. */
int rii_ivf_merge_top1_hdr_dev(const void *d_gathered, int G, int64_t B, const int64_t *d_counts, int64_t *d_out_ids, float *d_out_dists,
                               int64_t *d_out_counts, int32_t *d_out_any, void *stream);

/* ---- Multi-GPU behind the C ABI (NEW, round 4; not in the reference: it has no multi-device code) ----
 * One rii_comm per process and GPU = one RCCL communicator (over xGMI inside a node).  RCCL is bound at run time (dlopen of the
 * copy already in the process -- PyTorch-ROCm ships one -- else the system's librccl.so.1); without it rii_comm_init fails with
 * RII_ERR_HIP.  Rank 0 (or anyone) calls rii_comm_unique_id and hands the RII_COMM_ID_BYTES bytes to every rank by whatever means
 * the host program has (MPI, a file, torch.distributed); every rank then calls rii_comm_init (collective).  The sharded calls below
 * enqueue  engine kernels -> ONE ncclAllGather of pre-sized records -> unpack / merge kernel  on `stream` (NULL = the engine's);
 * the calls that use one communicator must be issued in the same order on every rank, one stream at a time, with the same
 * batch / topk / L / target-set arguments (every decision that leads to a collective is taken from those alone).
 * Failures (round 5): a rank whose OWN part of a call fails (its engine call, an allocation, a bad local argument) still takes part in
 * the batch's all-gather and returns its error afterwards; its peers return from the same call too -- query sharding: the failed
 * rank's rows read ids -1 / distances NaN / counts -1; database sharding: every record carries a 16-byte header {int64 id offset,
 * int32 status} and a non-zero status poisons the whole batch on every rank (ids -2, distances NaN), the top-k paths (which read one
 * word per batch anyway) also return RII_ERR_STATE -- so the ranks stay in step and the communicator stays usable.  The top-1 paths
 * are asynchronous and return RII_OK on the healthy ranks: THEIR callers recognise a poisoned batch by ids -2 (inverted index: also
 * counts -1).  Only when a collective ITSELF fails, a rank cannot allocate its exchange records, cannot write its record header or
 * cannot order its own stream (a device that is gone) is the communicator marked unusable -- the peers of that one call wait in
 * their all-gather; every later call on it returns RII_ERR_STATE; destroy it. */
#define RII_COMM_ID_BYTES 128
typedef struct rii_comm rii_comm;
int rii_comm_unique_id(void *id_out /* RII_COMM_ID_BYTES */);
int rii_comm_init(const void *id, int rank, int nranks, int device, rii_comm **out);
void rii_comm_destroy(rii_comm *c);
int rii_comm_rank(const rii_comm *c);
int rii_comm_size(const rii_comm *c);

/* Query sharding (index replicated on every GPU): every rank passes the SAME d_queries [B, D]; rank r answers rows
 * [r * (B / G) + min(r, B % G), ...) -- slices differ by at most one row -- and one all-gather of the packed rows gives every rank
 * all B rows: row b is exactly what rii_query_linear_dev / rii_query_ivf_dev return for query b (RiiCpp::QueryLinear, src/rii.h:195-242;
 * RiiCpp::QueryIvf, src/rii.h:244-326: its "stop at exactly L candidates in list order" rule is per query, so the inverted index is
 * sharded over queries).  Asynchronous on the stream. */
int rii_query_linear_qsharded_dev(rii_engine *e, rii_comm *c, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                                  int64_t S, int64_t *d_out_ids, float *d_out_dists, void *stream);
int rii_query_ivf_qsharded_dev(rii_engine *e, rii_comm *c, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                               int64_t S, int64_t L, int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts, void *stream);

/* The same for callers that run the collective themselves: rank r's record is [nmax * k] int64 ids, [nmax] int64 counts (only with
 * `counts`: the inverted index), [nmax * k] f32 distances, padded to rii_qshard_record_bytes() -- nmax = ceil(B / G) rows, of which the
 * rank fills its rii_qshard_begin(B, G, r + 1) - rii_qshard_begin(B, G, r) -- and rii_qshard_unpack_dev lays the G gathered records
 * out as the [B, k] outputs.  Stateless. */
int64_t rii_qshard_begin(int64_t B, int G, int rank);
int64_t rii_qshard_record_bytes(int64_t B, int G, int k, int counts);
int rii_qshard_unpack_dev(const void *d_gathered, int64_t B, int G, int k, int counts, int64_t *d_out_ids, float *d_out_dists,
                          int64_t *d_out_counts, void *stream);

/* Database sharding, linear search (Deep1B: 16 GB of codes -> 2 GB per GPU): rank r's engine holds the codes with global ids
 * [id_offset, id_offset + N_local) -- contiguous id ranges in rank order.  Every rank answers the WHOLE batch on its shard (k + 1
 * rows per query; one row for top-1), ONE all-gather, and every rank merges the G records under (distance, global id): the answer of
 * RiiCpp::QueryLinear (src/rii.h:195-242) on the concatenated database, GLOBAL ids, identical on every rank.  Where two of the merged
 * k + 1 best distances are bit-equal the reference's order is std::partial_sort's (src/rii.h:234-235) over all distances in index
 * order: those queries (d_out_tie [B] int32, or NULL) are replayed exactly with rii_linear_tie_emit_dev / _replay_dev's kernels and
 * one more all-gather (tie_cap rows per flagged query and rank, 0 = 12288; a longer list keeps the (distance, id) answer and is
 * marked in d_out_overflow [B] int32, or NULL).  top-1 is asynchronous on the stream; top-k synchronises once per batch (the host
 * reads one word: is any query flagged?).  d_tids_local: this rank's share of the target ids as LOCAL ids, S_local of them; S_global =
 * size of the whole target set, 0 = none (a rank may own none of the targets: S_local == 0, S_global != 0).
 * Any G, any topk (round 5): the shards' first ids travel in the record headers (nothing cached per communicator: two differently
 * sharded indices may share one); G * (topk + 1) > 8192 rows per query are merged in global scratch instead of LDS; the exact-tie
 * replay walks heaps of up to 1024 entries -- a flagged query with topk > 1024 keeps the (distance, id) order among its exactly tied
 * distances and is marked in d_out_overflow, like a candidate list above tie_cap. */
int rii_query_linear_dbsharded_dev(rii_engine *e, rii_comm *c, int64_t id_offset, const float *d_queries, int64_t B, int topk,
                                   const int64_t *d_tids_local, int64_t S_local, int64_t S_global, int64_t *d_out_ids,
                                   float *d_out_dists, int32_t *d_out_tie, int32_t *d_out_overflow, int tie_cap, void *stream);

/* Database sharding, inverted index: the protocol described above rii_ivf_list_lengths_dev, driven by the library -- the per-rank list
 * lengths all-gathered (nlist int32 per rank), the reference's global walk (src/rii.h:283-321) replayed on them by every rank, ONE
 * all-gather of the ranks' k + 1 best (position, global id, distance) rows merged under (distance, position), and the queries whose
 * k + 1 best distances tie exactly (d_out_tie [B] int32, or NULL) redone with every owned candidate gathered (rows = L) and
 * std::partial_sort (src/rii.h:312-313) replayed on the rebuilt sequence.  Coarse centres replicated (rii_set_coarse_centers on every
 * rank), posting lists over this rank's codes [id_offset, id_offset + N_local); N_global = codes of the whole database.  Outputs: GLOBAL
 * ids, distances, counts (topk, or 0 where the reference returns ({}, {})), identical on every rank.  top-1 is asynchronous on the
 * stream; top-k synchronises once per batch.  Any L <= N_global, any topk <= L, any G (round 5): see rii_query_ivf_shard_dev for the
 * shard kernel above L = 8192; the exact-tie replay gathers the flagged queries' candidates in groups of at most 512 MiB; topk + 1
 * rows beyond what a launch selects take the collect-all route (every candidate of every query gathered group by group, the replay is
 * the answer; d_out_tie is zero then). */
int rii_query_ivf_dbsharded_dev(rii_engine *e, rii_comm *c, int64_t id_offset, int64_t N_global, const float *d_queries, int64_t B,
                                int topk, const int64_t *d_tids_local, int64_t S_local, int64_t S_global, int64_t L,
                                int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts, int32_t *d_out_tie, void *stream);

/* Distance-table build alone (RiiCpp::DTable, src/rii.h:361-373) for B queries -> out[B,M,Ks] (host). */
int rii_dtable(rii_engine *e, const float *queries, int64_t B, float *out);
/* Coarse assignment alone (PQKMeans::predict_one over codes, src/rii.h:350-354): assign[n] in [0,nlist). */
int rii_assign(rii_engine *e, const uint8_t *codes, int64_t n, int32_t *assign);

/* Layout introspection (host only, no device needed): the filter scan of the M = 16 / 32 / 64, Ks = 256 shapes
 * (fscan_mx_kernel) splits the M table rows of a code over four lanes of a wave; this returns the subspace whose row lane
 * `lane` (0..63: lane 16 g + n works on code n of its group of 16) fetches as its lookup t (0 .. M/4 - 1), or -1 for bad
 * arguments.  The choice makes every LDS service group (16 lanes of a ds_read_b128 for M = 16 / 32, 32 lanes of a ds_read_b64
 * for M = 64) hit that many different bank slots (tests/test_capi.py checks both that and that the four lanes of a code cover
 * its M subspaces exactly once). */
int rii_fscan_lane_subspace(int M, int lane, int t);

/* Options (rii_set_option / rii_get_option; every one of them except "lut_mode" leaves the results bit-identical -- they select
 * among implementations that the parity tests compare with each other; defaults in brackets; measurements: INTEGRATION.md section 5):
 *   "lut_mode"         RII_LUT_EXACT [default] / RII_LUT_MFMA
 *   "scan_mode"        1 = 8-bit filter + exact re-rank [default], 0 = exhaustive fp32 scan only
 *   "scan_mx"          1 = the M = 16 / 32 / 64, Ks = 256 filter sums its table bytes on the matrix cores (fscan_mx_kernel) [default],
 *                      0 = on the vector ALU (fscan_kernel)
 *   "scan_pipe"        the top-1 scan of M = 16 (one tile per block) / 32 with unsigned table bytes judges a group's sums one group late,
 *                      behind the next group's matrix instructions: 1 = for batches of at most 128 queries [default], 2 = always, 0 = never
 *   "scan_dual"        M = 16: 1 = two 16-query tiles per scan block (fscan_mx_dual_kernel) [default], 0 = one
 *   "scan_order"       1 = the shapes without the rotated table layout scan an LDS-friendly permutation of the codes [default], 0 = id order
 *   "scan_chunks"      chunks of the code array per query tile (grid.x of the scan kernels); 0 = automatic [default]
 *   "fast_min_batch"   top-1 batches smaller than this take the exhaustive scan / the one-launch kernels [33]
 *   "cand_cap"         candidate slots per query of the filter stage (tests force small values to reach the overflow path) [automatic]
 *   "fused_tables"     1 = byte tables of M = 16 / 32, Ks = 256, Ds = 4 / 2 from ONE launch (qlut_fused_kernel), top-1 re-ranked from the
 *                      codebook [default]; 0 = the two-launch tile path
 *   "table_levels"     quantisation levels of those tables: 63, 127 [default] or 255
 *   "generic_table_levels" quantisation levels of the byte tables of every OTHER shape (any Ds: the Deep1B shape D = 96, M = 16 among
 *                      them) when the matrix-core scan of M = 16 / 32 reads them: 0 = automatic: 255 [default], or 63 / 127 / 255.  (Round 6: 63 levels sent
 *                      8 % of a structured 10 M-vector set through the candidate path -- 157 ms per 1024 queries against 7 with 255.)
 *   "fused_rerank"     1 = the top-1 re-rank runs as the tail of the scan launch (one launch per batch, one host flag per tile for
 *                      rii_query_linear_dev_to_host); 0 = rerank_top1_direct_kernel [default: measured equal or faster]
 *   "small_topk"       1 = a small batch over a small index (<= ~13 800 codes at M = 32) in ONE launch (small_topk_kernel) [default]
 *   "slice_topk"       1 = 1 - 8 queries on a larger index in ONE launch (slice_topk_kernel; exact ties redone by the general path:
 *                      decided by the host for host-pointer calls, by flag-gated tie kernels for asynchronous calls) [default]
 *   "host_spin"        1 = small host-pointer calls get their rows written into the engine's pinned block and wait on sequence flags
 *                      there instead of a D2H copy + stream synchronisation [default]
 *   "host_zero_copy"   host-pointer linear batches read their queries from / write their rows to the pinned block: 1 = up to 128 KiB
 *                      of queries [default], 2 = always, 0 = never
 *   "ivf_fused"        1 = one fused launch per batch for the inverted index with per-query exact fallback [default], 0 = the
 *                      std::partial_sort emulation kernels for every query
 *   "ivf_quad"         1 = top-1 batches of >= 768 queries over <= 1024 lists with w <= 7 (Ds = 4, Ks = 256, M = 16 / 32) run FOUR queries per
 *                      block with their tables interleaved in LDS (ivf_quad_kernel: one 16-byte read scores a centre for four queries)
 *                      [default], 2 = at every batch size and up to w = 32 (tests), 0 = one query per block (ivf_fused_kernel).
 *                      Identical results
 *   "ivf_rot"          1 = top-1 batches with L >= 2048 over <= 1024 unfiltered lists (Ks = 256, M = 64, Ds = 2 / 4) use the conflict-free
 *                      table gather (ivf_rot_kernel: table [ks][64 columns], lanes skewed in time over rotated 64-row tiles of the centres
 *                      and of the posting-order codes; + ~N*M bytes of device memory, rebuilt with the lists) [default], 2 = wherever the
 *                      kernel applies (M = 32 too: tests), 0 = off.  Identical results
 *   "ivf_inline_exact" 1 = a block of the fused kernel that flags its own query replays it itself [default], 0 = flag-gated exact
 *                      kernels behind every batch
 *   "ivf_list_codes"   1 = the fused kernel reads its candidates from a second copy of the codes kept in posting order (+N*M bytes of
 *                      device memory, rebuilt with the lists) [default], 0 = rows gathered by id.  Identical results
 *   "ivf_force_exact"  tests / measurement: 1 = every query of the fused path is flagged [0]
 *   "ivf_dbg_stop"     measurement only: ivf_quad_kernel / ivf_rot_kernel return after their table (1) / coarse (2) / selection (3)
 *                      phase (ivf_rot_kernel: 9 = at once) -- the rows are NOT answers then (tools/r5_ivf_phases.py, r6_rot_ab.py) [0]
 *   "shard_dbg_stop"   measurement only: ivf_shard_any_kernel (database-sharded inverted index) returns after its phase 1 .. 5 -- the rows
 *                      are NOT answers then (tools/r5_shard_phases*.sh) [0]
 *   "shard_pre"        1 = the database-sharded inverted index (rii_query_ivf_shard_dev / _dbsharded_dev) runs the coarse phase of a batch of
 *                      >= 256 queries over >= 4096 lists as a pre-pass with FOUR queries per block (shard_coarse_quad_kernel: tables interleaved [m][ks][query],
 *                      one 16-byte LDS read scores a centre for four queries; Ks = 256, M = 16 / 32, even Ds <= 8, w <= 7, coarse order
 *                      not in LDS) and the walk kernel starts from its picks [default], 2 = at every batch size, 0 = off.  Identical results
 *   "shard_force_replay" tests only: 1 = that kernel's fast coarse selection off, every query replays std::partial_sort over the coarse
 *                      distances.  Identical results [0]
 *   "lanes"            scratch-buffer sets: 2 [default] or 1 (see Threading below)
 *   "timing"           0 [default] / 1 (HIP events around every kernel) / 2 (only around the dominant kernel of a step): rii_timing_read
 * Read-only (rii_get_option): "lut_tile", "n_cu", "cand_total", "cand_max" (debug counters of the last filter pass; synchronise),
 *   "ivf_rot_launches", "ivf_quad_launches", "shard_pre_launches" (ivf_rot_kernel / ivf_quad_kernel / shard_coarse_quad_kernel launches so far: tests assert that the
 *   kernel under test really ran).
 *
 * Threading: every entry point locks the engine, concurrent callers are serialised.  The *_dev calls return after
 * enqueueing.  The engine keeps two sets of scratch buffers ("lanes"): a caller that issues successive batches alternately
 * on two streams gets one lane per stream and the batches overlap on the device (the table build, re-rank and launch gaps
 * of one hide behind the scan of the other: +10 % queries/s on the linear scan, +40-60 % on the inverted index and on
 * subset search); a call on a stream other than the one its lane served last is ordered behind that work with an event.
 * "lanes" = 1: one lane, calls on different streams serialise.  Index mutations, rii_timing_read and rii_synchronize wait
 * for both lanes.  The second lane's buffers are allocated on first use (single-stream callers never pay for them). */
int rii_set_option(rii_engine *e, const char *key, int64_t value);
int64_t rii_get_option(const rii_engine *e, const char *key);

/* Per-kernel HIP-event timing (option "timing" = 1: every kernel; 2: only the kernel that dominates a query step -- "scan",
 * "ivf_fused", "ivf_scan" -- two event records per step instead of ten): events are recorded on the launch stream
 * around every launch of the named kernel; reading synchronises the stream.
 * names: "lut", "scan", "ivf_coarse", "ivf_plan", "ivf_scan", "ivf_select", "assign", "gather", "select", "quant", "rerank",
 * "ivf_fused", "ivf_exact", "kth", "scan_order", "tie", "format", "ivf_shard", "shard_coarse". */
int rii_timing_read(rii_engine *e, const char *kernel, double *total_ms, int64_t *launches);
int rii_timing_reset(rii_engine *e);
int rii_synchronize(rii_engine *e);

#ifdef __cplusplus
}
#endif
#endif /* RII_AMD_H */
