// engine.hip -- host side of the engine: device-buffer ownership, batching/chunking policy, posting-list
// bookkeeping and the C ABI of include/rii_amd.h.  Mirrors rii::RiiCpp (src/rii.h:40-419) one-to-one; every
// arithmetic step happens in the HIP kernels of kernels.hip.  There is NO CPU fallback anywhere in this
// file: without a working HIP device rii_create() fails with RII_ERR_HIP.
#include "rii_amd.h"
#include "rii_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <thread>
#include <vector>

#define RII_API extern "C" __attribute__((visibility("default")))

using namespace riiamd;

static thread_local std::string g_err;
static int set_err(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return set_err(RII_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                               \
    } while (0)
#define RII_TRY(expr)                 \
    do {                              \
        int _r = (expr);              \
        if (_r != RII_OK) return _r;  \
    } while (0)

namespace riiamd {
thread_local LaunchEvents g_launch_events;
}

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // grow-only; contents are NOT preserved unless keep > 0 (bytes to carry over)
    int ensure(size_t bytes, size_t keep = 0, hipStream_t st = nullptr)
    {
        if (bytes <= cap) return RII_OK;
        size_t ncap = std::max(bytes, cap + cap / 2);
        void *np = nullptr;
        HIP_TRY(hipMalloc(&np, ncap));
        if (keep && p) {
            HIP_TRY(hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        if (p) HIP_TRY(hipFree(p));
        p = np;
        cap = ncap;
        return RII_OK;
    }
    void release()
    {
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace

// Everything one query call writes: scratch buffers plus the per-call state that describes their contents.  An engine
// holds two of these ("lanes"): the active one is the base sub-object of rii_engine, the other is parked.  A caller
// that alternates between two streams gets one lane per stream (begin_on swaps them), so the latency-bound stages of
// one batch (table build, re-rank, launch gaps) overlap the other batch's scan; single-stream callers never touch the
// second lane, whose buffers are only allocated on first use.
struct ScratchSet {
    DevBuf s_queries, s_tids, s_lut, s_best, s_out_ids, s_out_dists, s_out_counts, s_sub_codes, s_keys_a, s_keys_b,
        s_assign, s_coarse_d, s_coarse_i, s_cum, s_ncand, s_nvis, s_cand_i, s_cand_d, s_bitmap, s_fids, s_flen,
        s_hist, s_cnt, s_sample, s_qlut, s_slack, s_cand, s_cand_cnt, s_flag, s_segmin, s_thr16, s_gthr, s_qc, s_flag_list,
        s_tie_list, s_tie_hid, s_tie_hd, s_tie_chunk, s_lohi, s_fsub, s_ident, s_big, s_small_done, s_tile_done, s_tie_cnt2;
    bool have_ident = false;    // s_ident = [count | 0 .. 63]: work list of rii_linear_tie_emit_dev (every query of the call is "flagged")
    void *sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    void *h_pin = nullptr;          // pinned host staging for small batches (one H2D + one D2H per call)
    void *d_pin = nullptr;          // ... as the device sees it (coherent, mapped)
    size_t h_pin_cap = 0;
    DevBuf s_out_pack;              // [ids | counts | flags | dists] of a small batch, copied back in one transfer
    riiamd::IvfParams ivf_deferred; // host-pointer small batches: fallback kernels are launched only if a flag came back set
    bool ivf_has_deferred = false;
    int flag_parity = 0;        // which of the two flagged-query counters the next inverted-index launch group uses
    int tie_parity = 0;         // likewise for the asynchronous few-query linear path (s_tie_cnt2: two counters, zeroed by the replay kernel)
    bool tie_cnt_dirty = false; // a call failed between its slice kernel and its replay kernel: both counters are zeroed again first
    int lut_qt = 0;             // layout of the fp32 tables currently in s_lut: queries per interleaved tile (1 = plain)
    bool qlut_ready = false;    // the quantised tables of the current batch were produced by the fused table kernel
    int qlut_levels = 63;       // quantisation levels of the byte tables of the current batch (255: signed bytes, fscan_mx_* only)
    bool qlut_quarter = false;  // ... as quarter tables (qlut_fused_kernel): fscan_mx_kernel interleaves them while staging
    bool lut_valid = true;      // s_lut holds the exact fp32 tables of the current batch (false: only the byte tables were built)
    int64_t last_fs_B = 0;      // batch size of the last filter + re-rank call (debug counters)
    // `last_stream`/`order_ev` chain this lane's users when successive calls on it come from different streams
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t order_ev = nullptr;

    void release_all()
    {
        DevBuf *bufs[] = {&s_queries, &s_tids, &s_lut, &s_best, &s_out_ids, &s_out_dists, &s_out_counts, &s_sub_codes, &s_keys_a,
                          &s_keys_b, &s_assign, &s_coarse_d, &s_coarse_i, &s_cum, &s_ncand, &s_nvis, &s_cand_i, &s_cand_d,
                          &s_bitmap, &s_fids, &s_flen, &s_hist, &s_cnt, &s_sample, &s_qlut, &s_slack, &s_cand, &s_cand_cnt,
                          &s_flag, &s_segmin, &s_thr16, &s_gthr, &s_qc, &s_flag_list, &s_tie_list, &s_tie_hid, &s_tie_hd,
                          &s_tie_chunk, &s_lohi, &s_fsub, &s_out_pack, &s_ident, &s_big, &s_small_done, &s_tile_done, &s_tie_cnt2};
        for (DevBuf *b : bufs) b->release();
        if (sort_temp) (void) hipFree(sort_temp);
        sort_temp = nullptr; sort_temp_bytes = 0;
        if (h_pin) (void) hipHostFree(h_pin);
        h_pin = nullptr; h_pin_cap = 0;
        if (order_ev) (void) hipEventDestroy(order_ev);
        order_ev = nullptr;
        last_stream_valid = false;
        have_ident = false;
    }
};

struct rii_engine : ScratchSet {
    int M = 0, Ks = 0, Ds = 0, arch = RII_SIMD_AVX512, verbose = 0, device = 0;
    int QT = 4;
    int lut_mode = RII_LUT_EXACT;
    int scan_chunks = 0;        // 0 = auto
    int scan_mode = 1;          // 1 = 8-bit filter + exact re-rank for top-1 (fastscan.hip), 0 = exact scan only
    int fast_min_batch = 33;    // top-1 batches below this take the exact scan (option "fast_min_batch"; tools/sweep_fast_min.py)
    int cand_cap = 4096;        // candidate slots per query for the re-rank stage (lower bound; grows for small batches)
    DevBuf d_cand_peak;         // longest candidate list the top-1 re-rank has seen (device word) ...
    unsigned int *h_cand_peak = nullptr;   // ... and its pinned host copy, refreshed behind every batch
    bool cand_cap_forced = false;   // set by option "cand_cap" (tests force tiny buffers to reach the overflow path)
    int ivf_fused = 1;          // 1 = one fused launch for the common IVF case (exact fallback per query), 0 = off
    // option "table_levels": quantisation levels of the fused tables (qlut_fused_kernel).  Measured at the bench shape (tools/levels_ab.py,
    // profiles/r03_levels_ab.json): 63 -> 124 candidates per query, 127 -> 40, 255 -> 20; the scan itself is 3 % SLOWER with 255 (signed
    // bytes: more switching in the matrix core under a power-limited clock), so 127 -- non-negative bytes, no bias -- is the default.
    int table_levels = 127;
    int generic_levels = 0;     // option "generic_table_levels": 0 = automatic (255 on the matrix-core scans), else 63 / 127 / 255
    int shard_dbg_stop = 0;     // measurement only: ivf_shard_any_kernel returns after phase 1 .. 5 (wrong rows; tools/r5_shard_phases*.sh)
    int shard_pre = 1;          // option "shard_pre": the database-sharded inverted index runs its coarse phase as a pre-pass (shard_coarse_quad_kernel)
    int64_t shard_pre_launches = 0;
    int shard_force_replay = 0; // tests only: ivf_shard_any_kernel without its fast coarse selection (every query replays std::partial_sort)
    int ivf_dbg_stop = 0;       // measurement only: ivf_quad_kernel returns after phase 1 .. 3 (wrong rows; tools/r5_ivf_phases.py)
    int ivf_rot = 1;            // option "ivf_rot" (round 6): 1 = top-1 batches with L >= 2048 candidates over <= 1024 lists at M = 64: the conflict-free
                                // table gather (ivf_rot_kernel; at M = 32 its 64 KiB table costs more blocks per CU than the conflicts did:
                                // profiles/r06_rot_ab.json); 2 = wherever the kernel applies (tests); 0 = off
    int ivf_quad = 1;           // option "ivf_quad" (round 5): 1 = top-1 batches of >= 16 queries over <= 1024 lists run four queries per block (ivf_quad_kernel)
    int ivf_inline_exact = 1;   // option "ivf_inline_exact" (round 4): 1 = a block of ivf_fused_kernel that flags its query (tied coarse distances, tail
                                // walk, ties at the cut) replays it itself; 0 = the flag-gated exact kernels behind every batch (round 3)
    int ivf_list_codes = 1;     // option "ivf_list_codes" (round 4): 1 = a second copy of the codes in POSTING order (+N*M bytes) feeds the candidate phase of
                                // ivf_fused_kernel (contiguous rows per list, no dependent id load); 0 = rows gathered by id (round 3)
    int fused_tables = 1;       // option "fused_tables": 1 = qlut_fused_kernel + table-free top-1 re-rank (round 3), 0 = the two-launch tile path
    int ivf_force_exact = 0;    // tests: the fused kernel flags every query, so the exact LDS kernel answers all of them
    int timing = 0;
    hipStream_t stream = nullptr;
    int n_cu = 256;

    // host mirrors (the reference keeps everything in host memory: src/rii.h:77-82)
    std::vector<float> codewords;
    std::vector<uint8_t> codes;                      // N*M
    std::vector<uint8_t> centers;                    // nlist*M
    std::vector<std::vector<int32_t>> lists;         // posting lists
    int64_t N = 0;

    // device state
    DevBuf d_codewords, d_cnorm, d_codes, d_centers, d_symtab, d_pl_off, d_pl_ids, d_list_len;
    DevBuf d_lcodes; bool lcodes_valid = false;      // codes in posting order (option ivf_list_codes), rebuilt with the CSR
    bool lcodes_failed = false, rot_failed = false;  // the copy did not fit last time: not retried -- one failing hipMalloc per query otherwise --
                                                     // until the lists change or the option is set again
    // round 6 (option ivf_rot): centres and posting-order codes in rotated 64-row tiles for ivf_rot_kernel (every list on a tile boundary)
    DevBuf d_rcent, d_rlcodes, d_rl_toff; bool rot_valid = false;
    int64_t quad_launches = 0;  // ivf_quad_kernel launches so far (get_option "ivf_quad_launches")
    int64_t rot_launches = 0;   // ivf_rot_kernel launches so far (get_option "ivf_rot_launches": tests assert the kernel under test really ran)
    // LDS-friendly scan order of the filter stage (scanorder.hip): codes gathered in scan order + position -> id.
    // Windows of 1024 codes are independent, so appends only (re)order the windows past `scan_cov`.
    DevBuf d_scan_codes, d_scan_perm;
    // formatted lookups of the filter stage's conflict-free rotated layout (fastscan.hip: fs_rot_supported shapes): the code
    // bytes in fscan_mx_kernel's lane order (scan_mx = 1) or 2 bytes per code byte (fscan_kernel); codes are formatted
    // independently, so appends only format the tail past `fc_cov`
    DevBuf d_fcodes;

    int64_t fc_cov = 0;
    int scan_order = 1;         // option "scan_order"
    int scan_mx = 1;            // option "scan_mx": 1 = rotated shapes scan with fscan_mx_kernel (its own lookup order), 0 = fscan_kernel
    int slice_topk = 1;         // option "slice_topk": host-pointer calls of up to 8 queries on an index too large for the one-block
                                // kernel take ONE launch (slice_topk_kernel: slices on all CUs, last block merges, ties -> general path)
    int host_spin = 1;          // option "host_spin": small host-pointer calls answered by the one-launch kernel get their rows written
                                // straight into the pinned block and wait on a flag there instead of a D2H copy + stream synchronisation
    bool ivf_q_host = false;             // (set by host_query around the call: d_queries is the pinned host block, see IvfParams::q_host_off)
    unsigned int *spin_flag = nullptr;   // (set by host_query around the call: device address of the flags in the pinned block)
    unsigned int spin_seq = 0;
    bool spin_used = false;
    int64_t spin_nflags = 0;    // flag words the kernel that took the spin path raises (queries, or tiles of the fused re-rank)
    // option "host_zero_copy" (round 4): host-pointer linear batches hand the kernels the queries in the pinned block itself (the table
    // kernel and the re-rank read them over PCIe) and get their rows written straight into it: no H2D copy in front of the launches,
    // no D2H copy behind them.  1 (default): batches of at most 128 KiB of queries (measured at N = 1M, M = 32: B = 128 0.1055 ->
    // 0.0925 ms per call, B = 1024 0.372 vs 0.379: the 512 KiB go faster through the copy engine); 2: always; 0: never
    int host_zero_copy = 1;
    // option "fused_rerank" (round 4): top-1 of the M = 16 / 32, Ks = 256 filter re-ranked by the LAST chunk-block of every tile inside
    // the scan launch (fs_tail_rerank) instead of a second kernel.  Measured (tools/r4_measure.py, same box, N = 1M, M = 32): B = 1024
    // 0.3335 vs 0.3333 ms per step, B = 128 0.0833 vs 0.0798 -- the tail's chain of dependent round trips (arrival, counts, records,
    // codes, two batches of codeword gathers) is as long as the separate kernel's, which runs on B blocks instead of B / 16, and a
    // queued launch costs nothing -- so the default is 0.  1 buys ONE launch per batch with one host flag per tile behind the rows
    // (rii_query_linear_dev_to_host: B = 128 0.0901 vs 0.0938 ms per synchronous call).
    int fused_rerank = 0;
    int small_topk = 1;         // option "small_topk": a small batch over a small index in one launch after the tables (smalltopk.hip)
    // option "scan_pipe" (round 4): top-1 of M = 16 (one tile per block) / 32 with unsigned table bytes judges a group's sums one group late,
    // behind the next group's matrix instructions (fscan_mx_kernel<.., PIPE>: no `s_nop 7` between the last v_smfmac of a group and its
    // compares).  Same-box A/B (tools/opt_ab.py scan_pipe 0 1, profiles/r04_scan_pipe_ab.json): scan kernel B = 64 0.0551 -> 0.0540 ms, B = 128
    // 0.0602 -> 0.0576, B = 256 0.0916 -> 0.0914, B = 512 +0.3 %, B = 1024 0.3214 -> 0.3228 (+0.4 %), B = 4096 +0.5 %: with many tiles the other
    // waves of the SIMD already fill the matrix pipe's latency and the longer-lived accumulators cost a little.  1 = batches of at most 128
    // queries [default], 2 = always, 0 = never
    int scan_pipe = 1;
    int scan_dual = 1;          // option "scan_dual": M = 16 keeps two 16-query tiles per scan block (fscan_mx_dual_kernel)
    int64_t scan_cov = 0;       // codes [0, scan_cov) are final (whole windows)
    int64_t scan_N = -1;        // N the order was last completed for
    bool have_symtab = false, have_cnorm = false, lists_dirty = true;


    std::map<std::string, KernelTimer> timers;
    std::vector<hipEvent_t> event_pool;     // timing events are recycled (creating a pair per launch costs more than recording it)

    ScratchSet parked;          // the other lane (see ScratchSet)
    int lanes = 2;              // option "lanes": 1 = every call shares one scratch set (calls on different streams serialise)
    // every C-ABI entry point holds `mu` while it touches engine state or enqueues work (one caller at a time)
    mutable std::mutex mu;

};

namespace {

int64_t nlist_of(const rii_engine *e) { return e->M ? (int64_t) (e->centers.size() / (size_t) e->M) : 0; }

// option "timing": 1 = every kernel, 2 = only the kernels that dominate a query step (what bench.py keeps on inside its timed
// region: two event records per step instead of ten)
bool timer_wanted(const rii_engine *e, const char *name)
{
    if (e->timing == 1) return true;
    if (e->timing == 2) return !strcmp(name, "scan") || !strcmp(name, "ivf_fused") || !strcmp(name, "ivf_scan") || !strcmp(name, "ivf_shard") || !strcmp(name, "shard_coarse");
    return false;
}
hipEvent_t timer_event(rii_engine *e)
{
    if (!e->event_pool.empty()) {
        hipEvent_t ev = e->event_pool.back();
        e->event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

struct ScopedTimer {
    rii_engine *e;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    bool attach;
    // attach = true: the scope holds exactly ONE launch that goes through launch_timed(), which takes the two events with the
    // dispatch (no event packets in the stream)
    ScopedTimer(rii_engine *e_, const char *name_, hipStream_t st_, bool attach_ = false) : e(e_), st(st_), name(name_), attach(attach_)
    {
        if (e->timing && timer_wanted(e, name)) {
            a = timer_event(e);
            b = timer_event(e);
            if (!a || !b) { a = b = nullptr; return; }
            if (attach) { g_launch_events.start = a; g_launch_events.stop = b; }
            else (void) hipEventRecord(a, st);
        }
    }
    ~ScopedTimer()
    {
        if (a && b && attach) {
            if (g_launch_events.start == a) {          // the launch never happened (an error on the way): nothing to read
                g_launch_events = LaunchEvents();
                e->event_pool.push_back(a);
                e->event_pool.push_back(b);
                return;
            }
        }
        if (a && b) {
            if (!attach) (void) hipEventRecord(b, st);
            KernelTimer &t = e->timers[name];
            t.pending.emplace_back(a, b);
            // fold finished pairs into the totals so that an unread timer does not pile up events
            while (t.pending.size() > 64 && hipEventQuery(t.pending.front().second) == hipSuccess) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, t.pending.front().first, t.pending.front().second) == hipSuccess) {
                    t.total_ms += ms;
                    t.launches += 1;
                }
                e->event_pool.push_back(t.pending.front().first);
                e->event_pool.push_back(t.pending.front().second);
                t.pending.erase(t.pending.begin());
            }
        }
    }
};

int ensure_symtab(rii_engine *e)
{
    if (e->have_symtab) return RII_OK;
    RII_TRY(e->d_symtab.ensure((size_t) e->M * e->Ks * e->Ks * sizeof(float)));
    HIP_TRY(launch_symtab(e->d_codewords.as<float>(), e->M, e->Ks, e->Ds, e->arch, e->d_symtab.as<float>(),
                          e->stream));
    e->have_symtab = true;
    return RII_OK;
}

int upload_centers(rii_engine *e)
{
    if (e->centers.empty()) return RII_OK;
    RII_TRY(e->d_centers.ensure(e->centers.size()));
    HIP_TRY(hipMemcpyAsync(e->d_centers.p, e->centers.data(), e->centers.size(), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->rot_valid = false;
    return RII_OK;
}

// device CSR of the posting lists (rebuilt lazily after mutations)
int sync_lists(rii_engine *e)
{
    if (!e->lists_dirty) return RII_OK;
    const int64_t nlist = (int64_t) e->lists.size();
    std::vector<int64_t> off((size_t) nlist + 1, 0);
    std::vector<int32_t> len((size_t) nlist, 0);
    for (int64_t i = 0; i < nlist; ++i) {
        len[(size_t) i] = (int32_t) e->lists[(size_t) i].size();
        off[(size_t) i + 1] = off[(size_t) i] + (int64_t) e->lists[(size_t) i].size();
    }
    std::vector<int32_t> ids((size_t) std::max<int64_t>(off[(size_t) nlist], 1));
    for (int64_t i = 0; i < nlist; ++i)
        std::copy(e->lists[(size_t) i].begin(), e->lists[(size_t) i].end(), ids.begin() + off[(size_t) i]);
    RII_TRY(e->d_pl_off.ensure(off.size() * sizeof(int64_t)));
    RII_TRY(e->d_pl_ids.ensure(ids.size() * sizeof(int32_t)));
    RII_TRY(e->d_list_len.ensure(std::max<size_t>(len.size(), 1) * sizeof(int32_t)));
    HIP_TRY(hipMemcpyAsync(e->d_pl_off.p, off.data(), off.size() * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_pl_ids.p, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    if (!len.empty())
        HIP_TRY(hipMemcpyAsync(e->d_list_len.p, len.data(), len.size() * sizeof(int32_t), hipMemcpyHostToDevice,
                               e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->lists_dirty = false;
    e->lcodes_valid = false;
    e->rot_valid = false;
    e->lcodes_failed = false;
    e->rot_failed = false;
    return RII_OK;
}

// option ivf_list_codes: row pp of d_lcodes = the code of posting pp of the CSR id array (every id sits in exactly one list: N rows)
static int sync_list_codes_try(rii_engine *e, hipStream_t st)
{
    int64_t n = 0;
    for (const auto &l : e->lists) n += (int64_t) l.size();
    RII_TRY(e->d_lcodes.ensure((size_t) std::max<int64_t>(n, 1) * e->M));
    HIP_TRY(launch_gather_codes_i32(e->d_codes.as<uint8_t>(), e->M, e->d_pl_ids.as<int32_t>(), n, e->d_lcodes.as<uint8_t>(), st));
    HIP_TRY(hipStreamSynchronize(st));        // engine-level state (like the CSR itself): a call on the other lane's stream may read it next
    e->lcodes_valid = true;
    return RII_OK;
}
// (callers fall back to gathering rows by id: the copy is an optimisation.  A failure is remembered and the caller's error text kept)
int sync_list_codes(rii_engine *e, hipStream_t st)
{
    if (e->lcodes_valid) return RII_OK;
    if (e->lcodes_failed) return RII_ERR_HIP;
    const std::string keep = g_err;
    const int r = sync_list_codes_try(e, st);
    if (r != RII_OK) { e->lcodes_failed = true; g_err = keep; }
    return r;
}

// option ivf_rot: the centres and the posting-order codes in rotated tiles of 64 rows (ivf_rot_kernel, kernels.hip); list i starts at
// tile rl_toff[i] of d_rlcodes.  Built from d_centers / d_lcodes on the first query that wants them after the lists changed.
static int sync_rot_codes_try(rii_engine *e, hipStream_t st);
int sync_rot_codes(rii_engine *e, hipStream_t st)
{
    if (e->rot_valid) return RII_OK;
    if (e->rot_failed) return RII_ERR_HIP;
    const std::string keep = g_err;
    const int r = sync_rot_codes_try(e, st);
    if (r != RII_OK) { e->rot_failed = true; g_err = keep; }
    return r;
}
static int sync_rot_codes_try(rii_engine *e, hipStream_t st)
{
    RII_TRY(sync_list_codes(e, st));
    const int64_t nlist = (int64_t) e->lists.size();
    std::vector<int32_t> toff((size_t) nlist + 1, 0);
    for (int64_t i = 0; i < nlist; ++i) {
        const int64_t t = (int64_t) toff[(size_t) i] + ((int64_t) e->lists[(size_t) i].size() + 63) / 64;
        if (t > INT32_MAX) return set_err(RII_ERR_UNSUPPORTED, "rotated list tiles exceed int32");
        toff[(size_t) i + 1] = (int32_t) t;
    }
    const int64_t ntl = toff[(size_t) nlist], ntc = (nlist + 63) / 64;
    RII_TRY(e->d_rl_toff.ensure(toff.size() * sizeof(int32_t)));
    RII_TRY(e->d_rcent.ensure((size_t) std::max<int64_t>(ntc, 1) * 64 * e->M));
    RII_TRY(e->d_rlcodes.ensure((size_t) std::max<int64_t>(ntl, 1) * 64 * e->M));
    HIP_TRY(hipMemcpyAsync(e->d_rl_toff.p, toff.data(), toff.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(launch_rot_rows(e->d_centers.as<uint8_t>(), nlist, e->M, e->d_rcent.as<uint8_t>(), ntc, st));
    HIP_TRY(launch_rot_lists(e->d_lcodes.as<uint8_t>(), e->d_pl_off.as<int64_t>(), e->d_rl_toff.as<int32_t>(), (int) nlist, e->M,
                             e->d_rlcodes.as<uint8_t>(), ntl, st));
    HIP_TRY(hipStreamSynchronize(st));        // (toff is a local; engine-level state like the CSR)
    e->rot_valid = true;
    return RII_OK;
}

// coarse assignment of device-resident codes [d_codes, d_codes + num) -> host vector
int assign_device_codes(rii_engine *e, const uint8_t *d_codes, int64_t num, std::vector<int32_t> &out)
{
    out.resize((size_t) num);
    if (num == 0) return RII_OK;
    RII_TRY(ensure_symtab(e));
    RII_TRY(e->s_assign.ensure((size_t) num * sizeof(int32_t)));
    {
        ScopedTimer t(e, "assign", e->stream);
        HIP_TRY(launch_assign(d_codes, num, e->M, e->Ks, e->d_symtab.as<float>(), e->d_centers.as<uint8_t>(),
                              (int) nlist_of(e), e->s_assign.as<int32_t>(), e->stream));
    }
    HIP_TRY(hipMemcpyAsync(out.data(), e->s_assign.p, (size_t) num * sizeof(int32_t), hipMemcpyDeviceToHost,
                           e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RII_OK;
}

// RiiCpp::UpdatePostingLists, src/rii.h:335-359
int update_posting_lists(rii_engine *e, int64_t start, int64_t num)
{
    if (num == 0) return RII_OK;
    std::vector<int32_t> assign;
    RII_TRY(assign_device_codes(e, e->d_codes.as<uint8_t>() + (size_t) start * e->M, num, assign));
    for (int64_t n = 0; n < num; ++n) {
        const int32_t a = assign[(size_t) n];
        if (a < 0 || a >= (int32_t) e->lists.size())
            return set_err(RII_ERR_HIP, "coarse assignment produced an invalid list id %d", a);
        e->lists[(size_t) a].push_back((int32_t) (start + n));
    }
    e->lists_dirty = true;
    return RII_OK;
}

int append_codes(rii_engine *e, const uint8_t *codes, int64_t n)
{
    if (n == 0) return RII_OK;
    const size_t M = (size_t) e->M;
    const size_t old_bytes = (size_t) e->N * M, add = (size_t) n * M;
    if (e->N < e->scan_cov) e->scan_cov = 0;         // the code array was restarted (clear / set_state)
    if (e->N == 0 && e->h_cand_peak && e->d_cand_peak.p) {     // a new database: the candidate slots start from the default again
        *e->h_cand_peak = 0u;
        HIP_TRY(hipMemsetAsync(e->d_cand_peak.p, 0, 64, e->stream));
    }
    if (e->N < e->fc_cov) e->fc_cov = 0;
    e->scan_N = -1;
    e->codes.insert(e->codes.end(), codes, codes + add);
    RII_TRY(e->d_codes.ensure(old_bytes + add, old_bytes, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_codes.as<uint8_t>() + old_bytes, codes, add, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->N += n;
    return RII_OK;
}

// quantisation levels of the byte tables the generic (any Ds) quantisers produce: the matrix-core scans understand 127 / 255
// (255 whatever "table_levels" says for the fused tables: this path serves the shapes the fused kernel does not -- Ds other than 2 / 4,
// the Deep1B shape among them -- and there the step decides between a handful and thousands of candidates per query; 63 where only the
// byte-packed vector scan applies)
// (M = 16 / 32: the shapes whose scans the fused tables already drive with 127 / 255 levels; M = 64 keeps 63)
int generic_table_levels(const rii_engine *e)
{
    return (e->scan_mx && (e->M == 16 || e->M == 32) && fs_rot_supported(e->M, e->Ks, 1)) ? (e->generic_levels ? e->generic_levels : 255) : 63;
}

// need_fp32 = false: the caller only needs the byte tables of the filter (top-1 over a shape whose re-rank works from the codebook)
int build_lut(rii_engine *e, const float *d_queries, int64_t B, hipStream_t st, bool want_quant = false, int qt = 0,
              bool alloc_only = false, bool need_fp32 = true)
{
    if (qt <= 0) qt = e->QT;
    if (qt <= 0) qt = 1;                  // wide tables (widetab.hip): plain [b][M*Ks] in global memory
    const size_t tiles = (size_t) ((B + qt - 1) / qt);
    RII_TRY(e->s_lut.ensure(tiles * (size_t) e->M * e->Ks * qt * sizeof(float)));
    e->lut_qt = qt;
    if (alloc_only) return RII_OK;
    e->qlut_ready = false;
    e->qlut_quarter = false;
    e->qlut_levels = 63;
    e->lut_valid = true;
    if (want_quant && e->lut_mode == RII_LUT_EXACT && e->scan_mode == 1 && fastscan_supported(e->M, e->Ks)) {
        const int qr = fastscan_rows(e->M, e->Ks);
        RII_TRY(e->s_qlut.ensure((size_t) ((B + qr - 1) / qr) * e->M * e->Ks * qr));
        e->lut_qt = 1;                    // the fused kernel writes the plain [b][M*Ks] layout (coalesced; re-rank reads it)
        RII_TRY(e->s_slack.ensure((size_t) B * sizeof(int32_t)));
        RII_TRY(e->s_cand_cnt.ensure((size_t) B * sizeof(unsigned int)));
        RII_TRY(e->s_gthr.ensure((size_t) B * sizeof(uint32_t)));
        if (e->fused_tables && qlut_fused_supported(e->M, e->Ks, e->Ds, e->scan_mx)) {
            // one launch: exact entries in registers -> per-query step by a block-local reduction -> quarter tables (+ the fp32
            // table only when somebody reads it)
            const bool fp32 = need_fp32 || !rerank_direct_supported(e->M, e->Ks, e->Ds);
            RII_TRY(e->s_qlut.ensure(qlut_fused_bytes(B, e->M)));
            ScopedTimer t(e, "lut", st);
            HIP_TRY(launch_qlut_fused(d_queries, B, e->d_codewords.as<float>(), e->M, e->Ds, fp32 ? e->s_lut.as<float>() : nullptr,
                                      e->s_qlut.as<uint32_t>(), e->s_slack.as<int32_t>(), e->s_cand_cnt.as<unsigned int>(),
                                      e->s_gthr.as<uint32_t>(), e->table_levels, st));
            e->qlut_ready = true;
            e->qlut_quarter = true;
            e->qlut_levels = e->table_levels;
            e->lut_valid = fp32;
            return RII_OK;
        }
        if (lut_tile_supported(e->M, e->Ks, e->Ds, e->scan_mx)) {         // by tile: exact table, extrema, rotated byte rows, slack
            RII_TRY(e->s_lohi.ensure((size_t) B * e->M * 2 * sizeof(float)));
            ScopedTimer t(e, "lut", st);
            HIP_TRY(launch_lut_tile_build_quant(d_queries, B, e->d_codewords.as<float>(), e->M, e->Ds, e->s_lut.as<float>(),
                                                e->s_lohi.as<float>(), e->s_qlut.as<uint8_t>(), e->s_slack.as<int32_t>(),
                                                e->s_cand_cnt.as<unsigned int>(), e->s_gthr.as<uint32_t>(), st));
            e->qlut_ready = true;
            return RII_OK;
        }
        RII_TRY(e->s_qc.ensure((size_t) B * e->M * e->Ks));
        ScopedTimer t(e, "lut", st);
        // round 6: the matrix-core scans take the finer tables here too (option table_levels; the Deep1B shape, Ds = 6, comes this way)
        const int lv = generic_table_levels(e);
        HIP_TRY(launch_lut_build_quant(d_queries, B, e->d_codewords.as<float>(), e->M, e->Ks, e->Ds, e->arch,
                                       e->s_lut.as<float>(), e->s_qc.as<uint8_t>(), e->s_qlut.as<uint8_t>(),
                                       e->s_slack.as<int32_t>(),
                                       e->s_cand_cnt.as<unsigned int>(), e->s_gthr.as<uint32_t>(), e->scan_mx, st, lv));
        e->qlut_ready = true;            // ... and the candidate counters / shared thresholds are already reset
        e->qlut_levels = lv;
        return RII_OK;
    }
    ScopedTimer t(e, "lut", st);
    if (e->lut_mode == RII_LUT_MFMA) {
        if (!e->have_cnorm) {
            RII_TRY(e->d_cnorm.ensure((size_t) e->M * e->Ks * sizeof(float)));
            HIP_TRY(launch_codeword_norms(e->d_codewords.as<float>(), e->M, e->Ks, e->Ds, e->d_cnorm.as<float>(), st));
            HIP_TRY(hipStreamSynchronize(st));
            e->have_cnorm = true;
        }
        HIP_TRY(launch_lut_build_mfma(d_queries, B, e->d_codewords.as<float>(), e->d_cnorm.as<float>(), e->M, e->Ks,
                                      e->Ds, qt, e->s_lut.as<float>(), st));
    } else {
        HIP_TRY(launch_lut_build(d_queries, B, e->d_codewords.as<float>(), e->M, e->Ks, e->Ds, e->arch, qt,
                                 e->s_lut.as<float>(), st));
    }
    return RII_OK;
}

// queries per table tile of the exhaustive scan for a batch of B: 1 for a single query (see launch_scan_wk; two
// queries would walk the codes twice, which loses to one 4-query tile)
// top-1 over one / two queries: 4 / 8-byte table rows (scan_kernel<1 / 2>: HBM-bound on a large shard); else the engine's tile
int exact_tile_for(const rii_engine *e, int64_t B, int topk) { return (topk == 1 && B <= 2 && e->QT >= (int) B) ? (int) B : e->QT; }

void pick_chunks(const rii_engine *e, int64_t n_codes, int64_t B, int *chunks, int64_t *chunk_len, int qt = 0)
{
    if (qt <= 0) qt = e->QT;
    const int64_t tiles = (B + qt - 1) / qt;
    int64_t c = e->scan_chunks;
    if (c <= 0) {
        // ~2 workgroups' worth of tiles per CU, but never chunks shorter than 8K codes (table staging cost); the one / two-query
        // stream (scan_kernel<1 / 2>: two blocks resident per CU) runs best with four per CU (2 GB shard, B = 1: 512 chunks 4.57 TB/s,
        // 1024: 5.38, 2048: 5.19, 4096: 4.97 -- profiles/r04_deep125m_few_queries.json)
        const bool stream_form = qt <= 2 && (size_t) qt * e->M * e->Ks * sizeof(float) <= ((size_t) 64 << 10);     // (its static-table instances)
        const int64_t target = (stream_form ? 4LL : 2LL) * e->n_cu;
        c = (target + tiles - 1) / tiles;
        const int64_t max_c = std::max<int64_t>(1, n_codes / 8192);
        c = std::max<int64_t>(1, std::min(c, max_c));
        if (tiles >= e->n_cu) c = 1;
    }
    c = std::max<int64_t>(1, std::min<int64_t>(c, 65535));
    int64_t len = (n_codes + c - 1) / c;
    len = std::max<int64_t>(len, 1);
    *chunks = (int) ((n_codes + len - 1) / len);
    *chunk_len = len;
}

constexpr int64_t kScanOrderMinN = 1 << 16;     // below this the scan is too short for the order to matter

int ensure_scan_order(rii_engine *e, hipStream_t st)
{
    if (e->scan_N == e->N) return RII_OK;
    const size_t M = (size_t) e->M;
    const int64_t cov = std::min(e->scan_cov, e->N);
    RII_TRY(e->d_scan_codes.ensure((size_t) e->N * M, (size_t) cov * M, st));
    RII_TRY(e->d_scan_perm.ensure((size_t) e->N * sizeof(int32_t), (size_t) cov * sizeof(int32_t), st));
    {
        ScopedTimer t(e, "scan_order", st);
        HIP_TRY(launch_scan_order(e->d_codes.as<uint8_t>(), e->N, e->M, e->Ks, fastscan_rows(e->M, e->Ks), cov / 1024,
                                  e->d_scan_perm.as<int32_t>(),
                                  e->d_scan_codes.as<uint8_t>(), st));
    }
    e->scan_cov = (e->N / 1024) * 1024;
    e->scan_N = e->N;
    HIP_TRY(hipStreamSynchronize(st));      // index-side state: the other lane's stream may read it next
    return RII_OK;
}

int ensure_fcodes(rii_engine *e, hipStream_t st)
{
    if (e->fc_cov == e->N) return RII_OK;
    RII_TRY(e->d_fcodes.ensure((size_t) fcodes_bytes(e->N, e->M, e->scan_mx), (size_t) fcodes_bytes(e->fc_cov, e->M, e->scan_mx), st));
    {
        ScopedTimer t(e, "format", st);
        HIP_TRY(launch_fcodes_format(e->d_codes.as<uint8_t>(), nullptr, e->fc_cov, e->N, e->M, e->Ks, e->d_fcodes.as<uint16_t>(),
                                     e->scan_mx, st));
    }
    e->fc_cov = e->N;
    HIP_TRY(hipStreamSynchronize(st));      // index-side state: the other lane's stream may read it next
    return RII_OK;
}

// linear top-k (k > 1): the producers of the canonical (dist, id) result append the queries whose k+1 smallest distances
// hold an exact tie to s_tie_list ([0] = count, [1..] = query indices relative to b0); linear_tie_kernel replays
// std::partial_sort (src/rii.h:234-235) for them over the codes in the reference's index order.
int tie_list_reset(rii_engine *e, int64_t bc, hipStream_t st)
{
    RII_TRY(e->s_tie_list.ensure((size_t) (bc + 1) * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(e->s_tie_list.p, 0, sizeof(int32_t), st));
    return RII_OK;
}
// d_queries_direct != NULL: no table was built for the batch (the asynchronous few-query path): the chunk kernels build a flagged
// query's table themselves; only valid when every flagged query takes the chunked form (bc <= 16)
int tie_fixup(rii_engine *e, const uint8_t *d_codes_idx, int indirect, int64_t n_codes, int64_t b0, int64_t bc, int topk,
              const int64_t *d_remap, int64_t *d_out_ids, float *d_out_dists, hipStream_t st, const float *d_queries_direct = nullptr,
              int *d_nflag = nullptr, int *d_nflag_next = nullptr)
{
    if (!d_nflag) d_nflag = e->s_tie_list.as<int>();
    if (!linear_tie_supported(e->M, e->Ks))
        return set_err(RII_ERR_UNSUPPORTED, "M*Ks=%d tables do not fit LDS next to the tie-order work list", e->M * e->Ks);
    // the first few flagged queries: chunk-parallel distances + one wave replaying the heap (three small launches); any
    // further ones (tie-heavy data: every query flagged) one block each -- there all CUs are busy anyway
    int first = 0;
    if (linear_tie_chunked_supported(e->M, e->Ks, topk) && n_codes >= 4 * 8192) {
        int fq = (int) std::min<int64_t>(std::min<int64_t>(bc, 16), std::max<int64_t>(1, ((int64_t) 1 << 30) / (n_codes * 9)));
        RII_TRY(e->s_tie_chunk.ensure(linear_tie_chunked_scratch(n_codes, fq)));
        ScopedTimer t(e, "tie", st);
        HIP_TRY(launch_linear_tie_chunked(d_codes_idx, n_codes, e->M, e->Ks, e->s_lut.as<float>(), e->lut_qt, b0,
                                          e->s_tie_list.as<int32_t>() + 1, d_nflag, d_remap, d_out_ids, d_out_dists,
                                          topk, fq, e->s_tie_chunk.p, indirect, st, d_queries_direct,
                                          e->d_codewords.as<float>(), e->Ds, e->arch, d_nflag_next));
        first = fq;
        if (bc <= fq) return RII_OK;
    }
    if (d_queries_direct) return set_err(RII_ERR_STATE, "tie_fixup: table-free form needs the chunked kernels");
    const int grid = (int) std::min<int64_t>(bc - first, 2LL * e->n_cu);
    if (!linear_tie_heap_in_lds(e->M, e->Ks, topk)) {
        RII_TRY(e->s_tie_hid.ensure((size_t) grid * topk * sizeof(unsigned long long)));
    }
    ScopedTimer t(e, "tie", st);
    HIP_TRY(launch_linear_tie(d_codes_idx, n_codes, e->M, e->Ks, e->s_lut.as<float>(), e->lut_qt, b0,
                              e->s_tie_list.as<int32_t>() + 1, e->s_tie_list.as<int>(), d_remap, d_out_ids, d_out_dists,
                              topk, grid, e->s_tie_hid.as<unsigned long long>(), indirect, first, st));
    return RII_OK;
}

// subset search: the S target codes gathered once for the whole batch (plain layout), for the paths that scan plain codes
int gather_plain(rii_engine *e, const int64_t *d_tids, int64_t S, hipStream_t st, const uint8_t **out)
{
    RII_TRY(e->s_sub_codes.ensure((size_t) S * e->M));
    ScopedTimer t(e, "gather", st);
    HIP_TRY(launch_gather_codes(e->d_codes.as<uint8_t>(), e->M, d_tids, S, e->s_sub_codes.as<uint8_t>(), st));
    *out = e->s_sub_codes.as<uint8_t>();
    return RII_OK;
}

// arrival counters of the fused re-rank (one per scan tile): zeroed once, the last block of a tile puts the zero back
constexpr size_t kPinFlagWords = 1024;        // sequence flags at the head of a lane's pinned block (host_spin)
int ensure_tile_done(rii_engine *e, hipStream_t st)
{
    const size_t bytes = 1024 * sizeof(unsigned int);            // >= kMaxBatch / 16 tiles
    if (e->s_tile_done.cap < bytes) {
        RII_TRY(e->s_tile_done.ensure(bytes));
        HIP_TRY(hipMemsetAsync(e->s_tile_done.p, 0, bytes, st));
    }
    return RII_OK;
}

// the scan for B queries whose tables are in s_lut over the whole database (S == 0) or the S target ids d_remap (scored
// in the order given: position s stands for the code d_remap[s], and ids are translated through d_remap at the end)
int scan_topk(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_remap, int64_t S,
              int64_t *d_out_ids, float *d_out_dists, hipStream_t st)
{
    const int64_t n_codes = S ? S : e->N;
    const uint8_t *d_codes = S ? nullptr : e->d_codes.as<uint8_t>();     // plain codes in position order (gathered lazily)
    const uint8_t *d_codes_idx = nullptr;           // what the re-rank / tie-order kernels index: see `indirect`
    int indirect = 0;
    ScanParams sp;
    sp.n_codes = n_codes; sp.M = e->M; sp.Ks = e->Ks;
    sp.B = (int) B; sp.QT = e->QT; sp.best = nullptr; sp.keys = nullptr; sp.b0 = 0; sp.bc = 0;
    // small top-1 batches: the exact scan needs no candidate machinery and wins below ~128 queries (tools/sweep_batch.py)
    const bool small_top1 = (topk == 1 && B < e->fast_min_batch);
    if (e->scan_mode == 1 && !small_top1 && fastscan_supported(e->M, e->Ks) && topk <= rerank_topk_max_k()) {
        // stage 0: quantise the tables; stage 1: byte-table scan -> candidates; stage 2: exact re-rank
        const int qr = fastscan_rows(e->M, e->Ks);
        const int qpb = fscan_queries_per_block(e->M, e->Ks, e->scan_mx, e->scan_dual);     // queries one scan block answers
        const int64_t tiles = (B + qpb - 1) / qpb;                                          // scan blocks per chunk
        int64_t c = e->scan_chunks;
        if (c <= 0) {
            // one block per CU at a time (LDS), all blocks equally long: the scan takes ceil(tiles * c / n_cu) rounds, each
            // as long as n_codes / c codes plus a fixed cost per block (table staging, warm-up, candidate flush: worth
            // ~13K codes, tools/chunk_sweep.py).  Pick the c that minimises the product (63 tiles: 4 chunks = 252 blocks in
            // one round, not 5 = 315 in two; 188 tiles: 4 chunks = 3 rounds of a quarter, not one round of everything);
            // never chunks shorter than 8 slabs or more than 128 of them (every chunk starts from its own thresholds)
            // (at most 64: with one to three query tiles the product below keeps asking for more, shorter chunks, but every chunk block
            //  stages its tile's 128 KiB of tables and converges its own thresholds -- measured, N = 1M: B = 16 73.7 us with 122 chunks,
            //  58.1 with 64; B = 32 89.6 -> 71.7; B = 48 81.5 -> 72.4; tools/r4_small_batch_chunks.py)
            const int64_t cmax = std::max<int64_t>(1, std::min<int64_t>(64, n_codes / 8192));
            const double per_block = 13000.0;
            double best_cost = 1e300;
            c = 1;
            for (int64_t cc = 1; cc <= cmax; ++cc) {
                const double rounds = (double) ((tiles * cc + e->n_cu - 1) / e->n_cu);
                const double cost = rounds * ((double) n_codes / (double) cc + per_block);
                if (cost < best_cost) { best_cost = cost; c = cc; }
            }
        }
        c = std::max<int64_t>(1, std::min<int64_t>(c, 65535));
        int64_t len = std::max<int64_t>(1, (n_codes + c - 1) / c);
        len = (len + 1023) / 1024 * 1024;                                 // chunks start on a block-iteration boundary
        const int chunks = (int) ((n_codes + len - 1) / len);
        const int64_t G = (int64_t) chunks * fscan_segments_per_chunk(e->M, e->Ks, e->scan_mx);   // lane segments per query (top-k passes)
        // pass 1 of top-k only has to bound the k-th smallest quantised sum from above: a 1-in-`stride` sample of the
        // 1024-code slabs does (its k-th smallest is the ~(stride*k)-th smallest overall: a few more candidates, 1/stride
        // of the pass).  Needs enough sampled codes per lane segment for the bound to be tight and k distinct winners.
        const int64_t slabs = (len + 1023) / 1024;
        int stride = 1;
        // the looser bound costs candidates (~stride * k per query), and once most waves meet a candidate in every slab the
        // divergent emission path dominates pass 2: keep stride * k around a few hundred
        while (stride < 8 && slabs / (stride * 2) >= 8 && (int64_t) stride * 2 * topk <= 256 &&
               (n_codes / (stride * 2)) >= (int64_t) 64 * topk)
            stride *= 2;
        const bool topk_ok = topk == 1 || (int64_t) topk * 2 <= std::min<int64_t>(n_codes / stride, G);
        if (topk_ok) {
            // candidate slots per query: small batches are cut into many chunks (each with its own running minimum,
            // hence more candidates per query), and can afford far more slots: ~128 MiB of slots in total
            int cap = (int) std::min<int64_t>(262144, std::max<int64_t>(e->cand_cap, ((int64_t) 1 << 24) / std::max<int64_t>(B, 1)));
            // round 6: the longest candidate list of the batches before (a word the re-rank keeps, copied to pinned memory behind every
            // batch -- read here without a synchronisation, so possibly one batch old).  A list that outgrows its slots sends its query
            // through an exhaustive pass by ONE block (36 ms over 10 M codes); on a structured Deep-shaped set the lists of a 16-byte
            // code average 13 k entries with a long tail (profiles/r06_deep_structured_levels.json), so the slots follow what the data
            // needs: up to 4 GiB of them
            if (!e->h_cand_peak && topk == 1 && !e->cand_cap_forced) {       // (first filter pass of the engine)
                void *hp = nullptr;
                if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess && e->d_cand_peak.ensure(64) == RII_OK &&
                    hipMemsetAsync(e->d_cand_peak.p, 0, 64, st) == hipSuccess) {
                    e->h_cand_peak = static_cast<unsigned int *>(hp);
                    *e->h_cand_peak = 0u;
                } else {
                    if (hp) (void) hipHostFree(hp);
                    (void) hipGetLastError();
                }
            }
            if (e->h_cand_peak && topk == 1) {
                const int64_t peak = (int64_t) *reinterpret_cast<volatile unsigned int *>(e->h_cand_peak);
                const int64_t lim = std::min<int64_t>((int64_t) 1 << 20, ((int64_t) 1 << 29) / std::max<int64_t>(B, 1));
                if (peak > cap / 2) {
                    int64_t want = 1;
                    while (want < peak * 2) want <<= 1;
                    cap = (int) std::max<int64_t>(cap, std::min<int64_t>(lim, want));
                }
            }
            if (e->cand_cap_forced) cap = e->cand_cap;
            if (topk > 1) cap = std::max(cap, 16 * topk * stride);
            const int32_t *d_perm = nullptr;
            const uint8_t *d_scan = nullptr;          // what fscan_kernel reads
            if (fs_rot_supported(e->M, e->Ks, e->scan_mx)) {
                // conflict-free rotated layout: the scan reads formatted lookups, the exact stages index the database
                if (S) {
                    RII_TRY(e->s_fsub.ensure((size_t) fcodes_bytes(S, e->M, e->scan_mx)));
                    ScopedTimer t(e, "gather", st);
                    HIP_TRY(launch_fcodes_format(e->d_codes.as<uint8_t>(), d_remap, 0, S, e->M, e->Ks, e->s_fsub.as<uint16_t>(),
                                                 e->scan_mx, st));
                    d_scan = e->s_fsub.as<uint8_t>();
                    indirect = 1;
                } else {
                    RII_TRY(ensure_fcodes(e, st));
                    d_scan = e->d_fcodes.as<uint8_t>();
                }
                d_codes_idx = e->d_codes.as<uint8_t>();
            } else {
                if (S) RII_TRY(gather_plain(e, d_remap, S, st, &d_codes));
                d_scan = d_codes;
                d_codes_idx = d_codes;
                // whole-database scans run over the LDS-friendly copy of the codes; the re-rank maps positions back to ids
                if (e->scan_order && !S && n_codes >= kScanOrderMinN && scan_order_supported(e->M, e->Ks)) {
                    RII_TRY(ensure_scan_order(e, st));
                    d_scan = e->d_scan_codes.as<uint8_t>();
                    d_perm = e->d_scan_perm.as<int32_t>();
                }
            }
            const uint8_t *d_rr = d_perm ? d_scan : d_codes_idx;       // the re-rank reads the code at the scan position
            RII_TRY(e->s_qlut.ensure((size_t) ((B + qr - 1) / qr) * e->M * e->Ks * qr));
            RII_TRY(e->s_slack.ensure((size_t) B * sizeof(int32_t)));
            RII_TRY(e->s_cand.ensure((size_t) B * cap * sizeof(unsigned long long)));
            RII_TRY(e->s_cand_cnt.ensure((size_t) B * sizeof(unsigned int)));
            if (!e->qlut_ready) {
                ScopedTimer t(e, "quant", st);
                RII_TRY(e->s_qc.ensure((size_t) B * e->M * e->Ks));
                const int lv = generic_table_levels(e);
                HIP_TRY(launch_lut_quantize(e->s_lut.as<float>(), B, e->M, e->Ks, e->lut_qt, e->s_qc.as<uint8_t>(),
                                            e->s_qlut.as<uint8_t>(), e->s_slack.as<int32_t>(), e->scan_mx, st, lv));
                e->qlut_levels = lv;
            }
            if (!e->qlut_ready) HIP_TRY(hipMemsetAsync(e->s_cand_cnt.p, 0, (size_t) B * sizeof(unsigned int), st));
            e->last_fs_B = B;
            if (topk == 1) {
                RII_TRY(e->s_gthr.ensure((size_t) B * sizeof(uint32_t)));
                if (!e->qlut_ready)
                    HIP_TRY(hipMemsetAsync(e->s_gthr.p, 0xff, (size_t) B * sizeof(uint32_t), st));   // > any 16-bit threshold
                // no fp32 table was written (fused tables): the last chunk-block of every tile re-ranks the tile's queries straight from
                // the codebook inside the scan launch (round 4) -- one launch less per batch, rows possibly straight to the host
                const bool tail_rr = !e->lut_valid && e->fused_rerank && !d_perm && fscan_tail_supported(e->M, e->Ks, e->Ds, e->scan_mx);
                FsTail tl;
                if (tail_rr) {
                    RII_TRY(ensure_tile_done(e, st));
                    tl.queries = d_queries; tl.codewords = e->d_codewords.as<float>(); tl.codes = d_rr; tl.remap = d_remap;
                    tl.indirect = indirect; tl.Ds = e->Ds; tl.topk = topk; tl.out_ids = d_out_ids; tl.out_dists = d_out_dists;
                    tl.tile_done = e->s_tile_done.as<unsigned int>();
                    const int64_t nfl = fscan_tail_flags(e->M, e->Ks, e->scan_mx, e->scan_dual, B);
                    if (e->spin_flag && nfl * sizeof(unsigned int) <= kPinFlagWords * sizeof(unsigned int)) {
                        tl.host_flag = e->spin_flag; tl.seq = e->spin_seq;
                        e->spin_used = true; e->spin_nflags = nfl;
                    }
                }
                {
                    ScopedTimer t(e, "scan", st, true);
                    HIP_TRY(launch_fscan(d_scan, n_codes, e->M, e->Ks, e->s_qlut.as<uint8_t>(), e->s_slack.as<int32_t>(),
                                         (int) B, chunks, len, e->s_cand.as<unsigned long long>(),
                                         e->s_cand_cnt.as<unsigned int>(), cap, 0, nullptr, nullptr,
                                         e->s_gthr.as<uint32_t>(), 1, e->scan_mx, st, e->qlut_quarter ? 1 : 0, e->scan_dual, e->qlut_levels,
                                         tail_rr ? &tl : nullptr, e->scan_pipe == 2 || (e->scan_pipe == 1 && B <= 128)));
                }
                if (tail_rr) return RII_OK;
                ScopedTimer t(e, "rerank", st);
                if (!e->lut_valid) {     // no fp32 table was written: distances of the candidates straight from the codebook
                    HIP_TRY(launch_rerank_top1_direct(d_rr, n_codes, e->M, e->Ds, d_queries, e->d_codewords.as<float>(),
                                                      e->s_slack.as<int32_t>(), e->s_cand.as<unsigned long long>(),
                                                      e->s_cand_cnt.as<unsigned int>(), cap, d_remap, B, d_out_ids, d_out_dists, topk,
                                                      indirect, st));
                    return RII_OK;
                }
                HIP_TRY(launch_rerank_top1(d_rr, n_codes, e->M, e->Ks, e->s_lut.as<float>(), e->lut_qt,
                                           e->s_slack.as<int32_t>(), e->s_cand.as<unsigned long long>(),
                                           e->s_cand_cnt.as<unsigned int>(), cap, d_remap, d_perm, B, d_out_ids, d_out_dists, topk,
                                           indirect, st, e->d_cand_peak.as<unsigned int>()));
                if (e->h_cand_peak && e->d_cand_peak.p)
                    HIP_TRY(hipMemcpyAsync(e->h_cand_peak, e->d_cand_peak.p, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
                return RII_OK;
            }
            // top-k: pass 1 = per-lane-segment minima of the quantised sums, k-th smallest of them bounds the k-th
            // smallest sum from above; pass 2 = keep every code within the proven slack of that bound; exact re-rank.
            RII_TRY(e->s_segmin.ensure((size_t) B * G * sizeof(uint16_t)));
            RII_TRY(e->s_thr16.ensure((size_t) B * sizeof(uint32_t)));
            {
                ScopedTimer t(e, "scan", st, true);
                HIP_TRY(launch_fscan(d_scan, n_codes, e->M, e->Ks, e->s_qlut.as<uint8_t>(), e->s_slack.as<int32_t>(), (int) B,
                                     chunks, len, nullptr, nullptr, 0, 1, e->s_segmin.as<uint16_t>(), nullptr, nullptr, stride,
                                     e->scan_mx, st, e->qlut_quarter ? 1 : 0, e->scan_dual, e->qlut_levels));
            }
            {
                ScopedTimer t(e, "kth", st);
                HIP_TRY(launch_kth_threshold(e->s_segmin.as<uint16_t>(), G, B, topk, fastscan_max_sum(e->M, e->qlut_levels),
                                             e->s_slack.as<int32_t>(), e->s_thr16.as<uint32_t>(), st));
            }
            {
                ScopedTimer t(e, "scan", st, true);
                HIP_TRY(launch_fscan(d_scan, n_codes, e->M, e->Ks, e->s_qlut.as<uint8_t>(), e->s_slack.as<int32_t>(), (int) B,
                                     chunks, len, e->s_cand.as<unsigned long long>(), e->s_cand_cnt.as<unsigned int>(), cap,
                                     2, nullptr, e->s_thr16.as<uint32_t>(), nullptr, 1, e->scan_mx, st, e->qlut_quarter ? 1 : 0, e->scan_dual, e->qlut_levels));
            }
            RII_TRY(tie_list_reset(e, B, st));
            {
                ScopedTimer t(e, "rerank", st);
                HIP_TRY(launch_rerank_topk(d_rr, n_codes, e->M, e->Ks, e->s_lut.as<float>(), e->lut_qt,
                                           e->s_cand.as<unsigned long long>(), e->s_cand_cnt.as<unsigned int>(), cap, d_remap,
                                           d_perm, B, d_out_ids, d_out_dists, topk, e->s_tie_list.as<int32_t>() + 1,
                                           e->s_tie_list.as<int>(), indirect, st));
            }
            return tie_fixup(e, d_codes_idx, indirect, n_codes, 0, B, topk, d_remap, d_out_ids, d_out_dists, st);
        }
    }
    // the exhaustive kernels read the tile-interleaved layout: re-lay the tables out if they are in another one
    const int qt = exact_tile_for(e, B, topk);
    if (e->lut_qt != qt || !e->lut_valid) RII_TRY(build_lut(e, d_queries, B, st, false, qt));
    sp.lut = e->s_lut.as<float>();
    sp.QT = qt;
    if (S) RII_TRY(gather_plain(e, d_remap, S, st, &d_codes));
    sp.codes = d_codes;
    d_codes_idx = d_codes;
    if (topk == 1) {
        pick_chunks(e, n_codes, B, &sp.chunks, &sp.chunk_len, qt);
        // 16-byte table rows (4 queries per row) meet the same ds_read_b128 service groups as the filter's byte tables:
        // whole-database scans walk the LDS-friendly copy of the codes
        if (e->scan_order && qt == 4 && fastscan_rows(e->M, e->Ks) == 16 && !S && n_codes >= kScanOrderMinN &&
            scan_order_supported(e->M, e->Ks)) {
            RII_TRY(ensure_scan_order(e, st));
            sp.codes = e->d_scan_codes.as<uint8_t>();
            sp.perm = e->d_scan_perm.as<int32_t>();
            sp.chunk_len = (sp.chunk_len + 1023) / 1024 * 1024;          // chunks start on a 64-lane slab boundary
            sp.chunks = (int) ((n_codes + sp.chunk_len - 1) / sp.chunk_len);
        }
        RII_TRY(e->s_best.ensure((size_t) B * sizeof(unsigned long long)));
        sp.best = e->s_best.as<unsigned long long>();
        HIP_TRY(hipMemsetAsync(sp.best, 0xff, (size_t) B * sizeof(unsigned long long), st));
        {
            ScopedTimer t(e, "scan", st, true);
            HIP_TRY(launch_scan(sp, st));
        }
        HIP_TRY(launch_finalize_top1(sp.best, B, d_remap, d_out_ids, d_out_dists, topk, st));
        return RII_OK;
    }
    // general top-k: emit all packed keys for a chunk of queries, fully sort each row, keep the first topk.
    const int64_t budget_keys = (int64_t) 1 << 27;                 // 1 GiB per key buffer
    int64_t bc = std::max<int64_t>(e->QT, (budget_keys / std::max<int64_t>(n_codes, 1)) / e->QT * e->QT);
    bc = std::min<int64_t>(bc, (B + e->QT - 1) / e->QT * e->QT);
    while (bc > e->QT && bc * n_codes >= ((int64_t) 1 << 31)) bc -= e->QT;
    if (bc * n_codes >= ((int64_t) 1 << 31))
        return set_err(RII_ERR_UNSUPPORTED, "topk>1 over %lld codes exceeds the sort fallback's 2^31 key limit",
                       (long long) n_codes);
    RII_TRY(e->s_keys_a.ensure((size_t) bc * n_codes * sizeof(unsigned long long)));
    RII_TRY(e->s_keys_b.ensure((size_t) bc * n_codes * sizeof(unsigned long long)));
    for (int64_t b0 = 0; b0 < B; b0 += bc) {
        const int64_t cur = std::min<int64_t>(bc, B - b0);
        sp.keys = e->s_keys_a.as<unsigned long long>();
        sp.b0 = (int) b0; sp.bc = (int) cur;
        pick_chunks(e, n_codes, cur, &sp.chunks, &sp.chunk_len);
        {
            ScopedTimer t(e, "scan", st, true);
            HIP_TRY(launch_scan(sp, st));
        }
        {
            ScopedTimer t(e, "select", st);
            HIP_TRY(segmented_sort_keys(e->s_keys_a.as<unsigned long long>(), e->s_keys_b.as<unsigned long long>(),
                                        cur, n_codes, &e->sort_temp, &e->sort_temp_bytes, st));
            HIP_TRY(launch_gather_sorted_topk(e->s_keys_b.as<unsigned long long>(), cur, n_codes, topk, d_remap,
                                              d_out_ids + b0 * topk, d_out_dists + b0 * topk, st));
            RII_TRY(tie_list_reset(e, cur, st));
            HIP_TRY(launch_sorted_tie_flag(e->s_keys_b.as<unsigned long long>(), cur, n_codes, topk,
                                           e->s_tie_list.as<int32_t>() + 1, e->s_tie_list.as<int>(), st));
        }
        RII_TRY(tie_fixup(e, d_codes_idx, 0, n_codes, b0, cur, topk, d_remap, d_out_ids, d_out_dists, st));
    }
    return RII_OK;
}

int check_query_args(const rii_engine *e, int64_t B, int topk, int64_t S)
{
    if (B < 0) return set_err(RII_ERR_INVALID, "negative batch size");
    if (topk < 1 || (int64_t) topk > e->N)
        return set_err(RII_ERR_INVALID, "topk=%d must satisfy 1 <= topk <= N=%lld (src/rii.h:202)", topk,
                       (long long) e->N);
    if (S < 0 || S > e->N) return set_err(RII_ERR_INVALID, "S=%lld must satisfy S <= N (src/rii.h:220)", (long long) S);
    if (S != 0 && (int64_t) topk > S)
        return set_err(RII_ERR_INVALID, "topk=%d must be <= len(target_ids)=%lld (src/rii.h:219)", topk, (long long) S);
    return RII_OK;
}

constexpr int64_t kMaxBatch = 8192;       // queries per internal pass: bounds the table scratch (32 KiB fp32 per query)
constexpr int64_t kMaxBatchWide = 1024;   // ... of the wide-table shapes (up to 256 KiB per query)

// Linear search of a shape whose one-query table does not fit LDS (QT == 0; widetab.hip): exact fp32 tables in global memory,
// one packed key per (query, code), segmented sort -> canonical (dist, id) order, exact ties replayed over the key rows.
int query_linear_wide(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_remap, int64_t S,
                      int64_t *d_out_ids, float *d_out_dists, hipStream_t st)
{
    const int64_t n_codes = S ? S : e->N;
    RII_TRY(build_lut(e, d_queries, B, st, false, 1));
    const int64_t budget_keys = (int64_t) 1 << 27;                 // 1 GiB per key buffer
    int64_t bc = std::max<int64_t>(1, std::min<int64_t>(B, budget_keys / std::max<int64_t>(n_codes, 1)));
    while (bc > 1 && bc * n_codes >= ((int64_t) 1 << 31)) --bc;
    if (bc * n_codes >= ((int64_t) 1 << 31))
        return set_err(RII_ERR_UNSUPPORTED, "wide-table search over %lld codes exceeds the sort path's 2^31 key limit", (long long) n_codes);
    RII_TRY(e->s_keys_a.ensure((size_t) bc * n_codes * sizeof(unsigned long long)));
    RII_TRY(e->s_keys_b.ensure((size_t) bc * n_codes * sizeof(unsigned long long)));
    for (int64_t b0 = 0; b0 < B; b0 += bc) {
        const int64_t cur = std::min<int64_t>(bc, B - b0);
        {
            ScopedTimer t(e, "scan", st);
            HIP_TRY(launch_scan_wide(e->d_codes.as<uint8_t>(), n_codes, e->M, e->Ks, e->s_lut.as<float>(), d_remap, (int) b0, (int) cur,
                                     e->s_keys_a.as<unsigned long long>(), st));
        }
        ScopedTimer t(e, "select", st);
        HIP_TRY(segmented_sort_keys(e->s_keys_a.as<unsigned long long>(), e->s_keys_b.as<unsigned long long>(), cur, n_codes,
                                    &e->sort_temp, &e->sort_temp_bytes, st));
        HIP_TRY(launch_gather_sorted_topk(e->s_keys_b.as<unsigned long long>(), cur, n_codes, topk, d_remap, d_out_ids + b0 * topk,
                                          d_out_dists + b0 * topk, st));
        if (topk > 1) {
            RII_TRY(tie_list_reset(e, cur, st));
            HIP_TRY(launch_sorted_tie_flag(e->s_keys_b.as<unsigned long long>(), cur, n_codes, topk, e->s_tie_list.as<int32_t>() + 1,
                                           e->s_tie_list.as<int>(), st));
            HIP_TRY(launch_tie_rows(e->s_keys_a.as<unsigned long long>(), n_codes, b0, cur, e->s_tie_list.as<int32_t>() + 1,
                                    e->s_tie_list.as<int>(), topk, d_remap, d_out_ids, d_out_dists, st));
        }
    }
    return RII_OK;
}

// per-query arrival counters of the multi-block one-launch kernels: zeroed once, the kernels put the zeros back
static int ensure_small_done(rii_engine *e, hipStream_t st)
{
    if (e->s_small_done.cap < (size_t) kMaxBatch * sizeof(unsigned int)) {
        RII_TRY(e->s_small_done.ensure((size_t) kMaxBatch * sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(e->s_small_done.p, 0, (size_t) kMaxBatch * sizeof(unsigned int), st));
    }
    return RII_OK;
}

// the one-launch path of smalltopk.hip answers this call (host_query relies on the same predicate)
static bool takes_small_topk(const rii_engine *e, int64_t B, int topk, int64_t S)
{
    return e->QT != 0 && e->small_topk && B > 0 && B < e->fast_min_batch && B <= kMaxBatch &&
           small_topk_supported(e->M, e->Ks, e->Ds, S ? S : e->N, topk);
}

int query_linear_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S,
                     int64_t *d_out_ids, float *d_out_dists, hipStream_t st)
{
    if (B == 0) return RII_OK;
    const int64_t maxb = e->QT ? kMaxBatch : kMaxBatchWide;
    if (B > maxb) {
        const int64_t D = (int64_t) e->M * e->Ds;
        for (int64_t b0 = 0; b0 < B; b0 += maxb) {
            const int64_t cur = std::min<int64_t>(maxb, B - b0);
            RII_TRY(query_linear_dev(e, d_queries + b0 * D, cur, topk, d_tids, S, d_out_ids + b0 * topk,
                                     d_out_dists + b0 * topk, st));
        }
        return RII_OK;
    }
    if (e->QT == 0) return query_linear_wide(e, d_queries, B, topk, S ? d_tids : nullptr, S, d_out_ids, d_out_dists, st);
    {
        // small index, small batch (the reference's README pattern: one query per call, N ~ 10^4, topk = 3): the exact
        // distances of all codes of a query fit LDS -- tables, selection and the tie order in ONE launch (smalltopk.hip)
        const int64_t n_codes = S ? S : e->N;
        if (takes_small_topk(e, B, topk, S)) {
            const bool own_tables = e->lut_mode == RII_LUT_EXACT;       // matrix-core tables come from their own kernel
            if (!own_tables) RII_TRY(build_lut(e, d_queries, B, st, false, 1));
            e->lut_valid = false;                                        // s_lut does not hold this batch's tables
            RII_TRY(e->s_keys_a.ensure(std::max<size_t>(small_topk_scratch(n_codes, B), 16)));
            RII_TRY(ensure_small_done(e, st));
            ScopedTimer t(e, "scan", st);
            HIP_TRY(launch_small_topk(e->d_codes.as<uint8_t>(), n_codes, e->M, e->Ks, own_tables ? nullptr : e->s_lut.as<float>(), d_queries,
                                      e->d_codewords.as<float>(), e->Ds, e->arch, B, topk, S ? d_tids : nullptr,
                                      e->s_keys_a.as<unsigned long long>(), e->s_small_done.as<unsigned int>(), d_out_ids, d_out_dists, st,
                                      e->spin_flag, e->spin_seq));
            e->spin_used = e->spin_flag != nullptr;
            e->spin_nflags = B;
            return RII_OK;
        }
    }
    {
        // a few queries per ASYNCHRONOUS call on a large index, topk > 1 (round 4): the one-launch slice kernel (smalltopk.hip) with a
        // DEVICE-side tie fallback -- the kernel appends the queries whose k + 1 smallest distances tie to the tie list, and the
        // flag-gated tie kernels behind it (tieorder.hip: they read the list's length on the device) redo exactly those in
        // std::partial_sort's order; no host decision, so the call stays asynchronous.  (Host-pointer calls keep the host's
        // decision: host_query.  top-1 needs no fallback at all and keeps the exhaustive scan, which is as fast at that size.)
        const int64_t n_codes = S ? S : e->N;
        if (topk > 1 && B <= 8 && e->slice_topk && e->QT != 0 && e->lut_mode == RII_LUT_EXACT && linear_tie_supported(e->M, e->Ks) &&
            slice_topk_supported(e->M, e->Ks, e->Ds, n_codes, B, topk)) {
            RII_TRY(e->s_keys_b.ensure(std::max<size_t>(slice_topk_scratch(n_codes, B, topk), 16)));
            RII_TRY(e->s_flag.ensure(8 * sizeof(int32_t)));
            RII_TRY(ensure_small_done(e, st));
            // four launches per call and nothing else (round 4; was: counter memset, slice kernel, table kernel, three tie kernels): the count
            // of flagged queries alternates between two words, each zeroed by the replay kernel of the call in front of its next use,
            // and a flagged query's exact table is built by the tie kernels' own blocks
            const bool direct = linear_tie_chunked_supported(e->M, e->Ks, topk) && n_codes >= 4 * 8192;    // (tie_fixup's chunked form)
            if (!direct) {
                RII_TRY(tie_list_reset(e, B, st));
                {
                    ScopedTimer t(e, "scan", st);
                    HIP_TRY(launch_slice_topk(e->d_codes.as<uint8_t>(), n_codes, e->M, e->Ks, d_queries, e->d_codewords.as<float>(), e->Ds, e->arch, B, topk,
                                              S ? d_tids : nullptr, e->s_keys_b.as<unsigned long long>(), e->s_small_done.as<unsigned int>(), d_out_ids,
                                              d_out_dists, e->s_flag.as<int32_t>(), st, nullptr, 0, e->s_tie_list.as<int32_t>() + 1, e->s_tie_list.as<int>()));
                }
                RII_TRY(build_lut(e, d_queries, B, st, false, 0));                  // the tie kernels' exact tables (B <= 8: microseconds)
                return tie_fixup(e, e->d_codes.as<uint8_t>(), S ? 1 : 0, n_codes, 0, B, topk, S ? d_tids : nullptr, d_out_ids, d_out_dists, st);
            }
            RII_TRY(e->s_tie_list.ensure((size_t) (B + 1) * sizeof(int32_t)));
            if (!e->s_tie_cnt2.p || e->tie_cnt_dirty) {
                RII_TRY(e->s_tie_cnt2.ensure(2 * sizeof(int)));
                HIP_TRY(hipMemsetAsync(e->s_tie_cnt2.p, 0, 2 * sizeof(int), st));
                e->tie_parity = 0;
                e->tie_cnt_dirty = false;
            }
            int *cnt = e->s_tie_cnt2.as<int>() + e->tie_parity, *cnt_next = e->s_tie_cnt2.as<int>() + (e->tie_parity ^ 1);
            e->tie_parity ^= 1;
            e->tie_cnt_dirty = true;                                                 // until the replay kernel of this call is enqueued
            {
                ScopedTimer t(e, "scan", st);
                HIP_TRY(launch_slice_topk(e->d_codes.as<uint8_t>(), n_codes, e->M, e->Ks, d_queries, e->d_codewords.as<float>(), e->Ds, e->arch, B, topk,
                                          S ? d_tids : nullptr, e->s_keys_b.as<unsigned long long>(), e->s_small_done.as<unsigned int>(), d_out_ids,
                                          d_out_dists, e->s_flag.as<int32_t>(), st, nullptr, 0, e->s_tie_list.as<int32_t>() + 1, cnt));
            }
            e->lut_valid = false;                                                    // (no table of this batch exists)
            RII_TRY(tie_fixup(e, e->d_codes.as<uint8_t>(), S ? 1 : 0, n_codes, 0, B, topk, S ? d_tids : nullptr, d_out_ids, d_out_dists, st,
                              d_queries, cnt, cnt_next));
            e->tie_cnt_dirty = false;
            return RII_OK;
        }
    }
    {
        const bool small_top1 = (topk == 1 && B < e->fast_min_batch);
        RII_TRY(build_lut(e, d_queries, B, st, !small_top1, small_top1 ? exact_tile_for(e, B, topk) : 0, false, /*need_fp32=*/topk > 1));
    }
    return scan_topk(e, d_queries, B, topk, S ? d_tids : nullptr, S, d_out_ids, d_out_dists, st);
}

// target_ids: one bitmap per batch + order-preserving compaction of every posting list (s_fids / s_flen), replacing the
// per-posting std::binary_search of src/rii.h:294
int filter_lists_by_targets(rii_engine *e, const int64_t *d_tids, int64_t S, hipStream_t st)
{
    const int64_t nlist = nlist_of(e);
    const size_t words = (size_t) ((e->N + 31) / 32);
    RII_TRY(e->s_bitmap.ensure(words * sizeof(uint32_t)));
    RII_TRY(e->s_fids.ensure((size_t) std::max<int64_t>(e->N, 1) * sizeof(int32_t)));
    RII_TRY(e->s_flen.ensure((size_t) nlist * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(e->s_bitmap.p, 0, words * sizeof(uint32_t), st));
    HIP_TRY(launch_bitmap_set(d_tids, S, e->s_bitmap.as<uint32_t>(), st));
    HIP_TRY(launch_filter_lists(e->d_pl_off.as<int64_t>(), e->d_pl_ids.as<int32_t>(), (int) nlist, e->s_bitmap.as<uint32_t>(),
                                e->s_fids.as<int32_t>(), e->s_flen.as<int32_t>(), st));
    return RII_OK;
}

// scratch of ivf_exact_big_kernel: one slice per block of its persistent grid, at most ~256 MiB in total
int ensure_big_scratch(rii_engine *e, int nlist, int64_t L, int64_t B, int *grid)
{
    const size_t per = ivf_exact_big_scratch(nlist, L);
    int64_t g = std::min<int64_t>(std::min<int64_t>(B, 256), std::max<int64_t>(1, ((int64_t) 256 << 20) / (int64_t) per));
    *grid = (int) g;
    return e->s_big.ensure(per * (size_t) g);
}

int query_ivf_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S,
                  int64_t L, int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts, hipStream_t st,
                  int32_t *d_flag_defer = nullptr)
{
    e->ivf_has_deferred = false;
    if (B == 0) return RII_OK;
    const int64_t maxb = e->QT ? kMaxBatch : kMaxBatchWide;
    if (B > maxb) {
        const int64_t D = (int64_t) e->M * e->Ds;
        for (int64_t b0 = 0; b0 < B; b0 += maxb) {
            const int64_t cur = std::min<int64_t>(maxb, B - b0);
            RII_TRY(query_ivf_dev(e, d_queries + b0 * D, cur, topk, d_tids, S, L, d_out_ids + b0 * topk,
                                  d_out_dists + b0 * topk, d_out_counts + b0, st));
        }
        return RII_OK;
    }
    const int64_t nlist = nlist_of(e);
    RII_TRY(sync_lists(e));

    IvfParams p;
    p.codes = e->d_codes.as<uint8_t>(); p.N = e->N; p.M = e->M; p.Ks = e->Ks;
    p.QT = 1;                     // inverted-index kernels stage ONE query's table per block: plain [b][M*Ks] layout
    p.queries = nullptr; p.codewords = e->d_codewords.as<float>(); p.Ds = e->Ds; p.arch = e->arch;
    p.centers = e->d_centers.as<uint8_t>(); p.nlist = (int) nlist;
    p.pl_off = e->d_pl_off.as<int64_t>();
    p.pl_ids = e->d_pl_ids.as<int32_t>();
    p.list_len = e->d_list_len.as<int32_t>();
    p.topk = topk; p.L = L;
    // w of src/rii.h:266-277
    double wd = (S == 0) ? std::round((double) L * (double) nlist / (double) e->N)
                         : std::round((double) L * (double) nlist / (double) S);
    int64_t w = (int64_t) (size_t) wd + 3;
    if (nlist < w) w = nlist;
    p.w = w;

    p.lcodes = nullptr;
    if (S == 0 && e->ivf_list_codes && e->ivf_fused) {      // unfiltered: the fused kernel reads its candidates' rows in posting order
        // (a second N * M bytes of HBM, built on the first unfiltered query after the lists change; where it does not fit the kernel
        //  gathers the rows by id as before -- the copy is an optimisation, never a precondition)
        if (sync_list_codes(e, st) == RII_OK) p.lcodes = e->d_lcodes.as<uint8_t>();
        else (void) hipGetLastError();
    }
    if (S != 0) {       // order-preserving filter of every list by the batch's target ids
        RII_TRY(filter_lists_by_targets(e, d_tids, S, st));
        p.pl_ids = e->s_fids.as<int32_t>();
        p.list_len = e->s_flen.as<int32_t>();
    }

    // chunk the batch so that the candidate scratch stays below ~1 GiB
    const int64_t stride = (topk == 1) ? 1 : L;
    int64_t bc = B;
    if (topk != 1) bc = std::max<int64_t>(1, std::min<int64_t>(B, ((int64_t) 1 << 27) / std::max<int64_t>(L, 1)));
    RII_TRY(e->s_coarse_d.ensure((size_t) bc * nlist * sizeof(float)));
    RII_TRY(e->s_coarse_i.ensure((size_t) bc * nlist * sizeof(int32_t)));
    RII_TRY(e->s_cum.ensure((size_t) bc * (nlist + 1) * sizeof(int32_t)));
    RII_TRY(e->s_ncand.ensure((size_t) bc * sizeof(int32_t)));
    RII_TRY(e->s_nvis.ensure((size_t) bc * sizeof(int32_t)));
    RII_TRY(e->s_cand_i.ensure((size_t) bc * stride * sizeof(int32_t)));
    RII_TRY(e->s_cand_d.ensure((size_t) bc * stride * sizeof(float)));
    p.coarse_dist = e->s_coarse_d.as<float>(); p.coarse_id = e->s_coarse_i.as<int32_t>();
    p.cum = e->s_cum.as<int32_t>(); p.ncand = e->s_ncand.as<int32_t>(); p.nvis = e->s_nvis.as<int32_t>();
    p.cand_id = e->s_cand_i.as<int32_t>(); p.cand_dist = e->s_cand_d.as<float>(); p.cand_stride = stride;
    const bool fused = e->ivf_fused && ivf_fused_supported(e->M, e->Ks, (int) nlist, w, topk);
    // a shape outside the fused kernel (w > 32 next to a large nlist, topk >= 1023): every query through the big exact kernel;
    // option ivf_fused = 0 keeps selecting the one-lane emulation kernels (tests)
    // (a shape whose table does not fit LDS at all -- QT == 0, widetab.hip -- has no other inverted-index kernel)
    const bool big_all = !fused && (e->ivf_fused || e->QT == 0) && ivf_exact_big_supported(e->M, e->Ks, w, topk);
    RII_TRY(e->s_flag.ensure((size_t) bc * sizeof(int32_t)));
    p.flag = fused ? e->s_flag.as<int32_t>() : nullptr;
    // round 4 (option ivf_inline_exact): a block of the fused kernel that flags its query redoes it itself (one global scratch slice
    // per query of the launch group, at most 256 MiB in all) -- no flag-gated second launch behind every batch
    const size_t inl_per_q = ivf_exact_big_scratch((int) nlist, L);
    const bool inline_exact = fused && e->ivf_inline_exact && e->QT != 0 && ivf_exact_big_supported(e->M, e->Ks, w, topk) &&
                              inl_per_q * (size_t) bc <= ((size_t) 256 << 20);
    if (inline_exact) {
        RII_TRY(e->s_big.ensure(inl_per_q * (size_t) bc));
        p.inl_scratch = e->s_big.as<unsigned char>();
        p.inl_per_q = inl_per_q;
    }
    const bool defer = fused && d_flag_defer && bc >= B;       // one launch group: the caller inspects the flags itself
    if (defer) p.flag = d_flag_defer;
    p.sel_cap = ivf_fused_sel_cap((int) nlist, w);
    p.force_flag = e->ivf_force_exact;
    p.flag_list = nullptr; p.nflag = nullptr;
    if (fused) {
        // [0], [1] = two counters used by alternate launch groups, [2..] = flagged query indices of the current group.  The
        // fused kernel zeroes the OTHER counter (its last reader, the previous group's exact kernel, is behind it in stream
        // order), so no memset sits in front of every batch.
        void *before = e->s_flag_list.p;
        RII_TRY(e->s_flag_list.ensure((size_t) (bc + 2) * sizeof(int32_t)));
        if (e->s_flag_list.p != before) {
            HIP_TRY(hipMemsetAsync(e->s_flag_list.p, 0, 2 * sizeof(int32_t), st));
            e->flag_parity = 0;
        }
        p.flag_list = e->s_flag_list.as<int32_t>() + 2;
    }
    if (fused && e->lut_mode == RII_LUT_EXACT) {
        RII_TRY(build_lut(e, d_queries, B, st, false, 1, /*alloc_only=*/true));   // tables are built inside the fused kernel
        p.queries = d_queries;
        p.q_host_off = (e->ivf_q_host && e->Ds == 4 && e->Ks == 256) ? 1 : 0;      // (host_query: the query sits in the pinned block)
    } else {
        RII_TRY(build_lut(e, d_queries, B, st, false, 1));
    }
    p.lut = e->s_lut.as<float>();

    for (int64_t b0 = 0; b0 < B; b0 += bc) {
        p.B = std::min<int64_t>(bc, B - b0);
        p.b0 = (int) b0;
        p.out_ids = d_out_ids + b0 * topk;
        p.out_dists = d_out_dists + b0 * topk;
        p.out_counts = d_out_counts + b0;
        if (fused) {
            p.nflag = e->s_flag_list.as<int>() + e->flag_parity;
            p.nflag_next = e->s_flag_list.as<int>() + (e->flag_parity ^ 1);
            e->flag_parity ^= 1;
            // common case answered in one launch; queries whose answer could hinge on std::partial_sort's internal
            // order raise flag[b] and are redone by the exact emulation kernels below (which skip the others)
            if (defer && e->spin_flag) { p.host_flag = e->spin_flag; p.host_seq = e->spin_seq; }
            // round 5: four queries per block (tables interleaved: one 16-byte LDS read scores a centre for four queries) for the
            // batched top-1 case; one-query calls, host-resident queries / flags and the other shapes keep the one-query blocks
            // (from 768 queries: below that one-query blocks have CUs -- and each CU's LDS -- to themselves and finish their 16 us chain
            //  before the 21 us chain of a four-query block; measured 16 / 64 / ... / 4096 queries: tools/r5_ivf_phases.py)
            // (and up to w = 7: the selection's rounds are serial per block -- w = 13, the subset search of configs[3], measured 38.7 us
            //  against 35.0 for one-query blocks)
            const bool quad = e->ivf_quad && p.queries && !p.q_host_off && !p.host_flag && (e->ivf_quad > 1 || (p.B >= 768 && w <= 7)) &&
                              ivf_quad_supported(e->M, e->Ks, e->Ds, (int) nlist, w, topk);
            p.kcap = quad ? e->ivf_dbg_stop : 0;
            // round 6: the conflict-free table gather where the candidate phase carries the kernel (L >= 2048: the reference's own
            // harness runs L = 5000 at M = 64); needs the rotated tile copies, so unfiltered lists only
            bool rot = !quad && e->ivf_rot && p.lcodes && p.queries && !p.q_host_off && !p.host_flag && (e->ivf_rot > 1 || (L >= 2048 && e->M == 64)) &&
                       ivf_rot_supported(e->M, e->Ks, e->Ds, (int) nlist, w, topk) && ivf_rot_fits(L, w);
            if (rot) {
                if (sync_rot_codes(e, st) == RII_OK) {
                    p.rcent = e->d_rcent.as<uint8_t>(); p.rlcodes = e->d_rlcodes.as<uint8_t>(); p.rl_toff = e->d_rl_toff.as<int32_t>();
                } else { (void) hipGetLastError(); rot = false; }               // (the copies are an optimisation, never a precondition)
            }
            if (rot) p.kcap = e->ivf_dbg_stop;
            ScopedTimer t(e, "ivf_fused", st, true);
            HIP_TRY(quad ? launch_ivf_quad(p, st) : rot ? launch_ivf_rot(p, st) : launch_ivf_fused(p, st));
            if (rot) e->rot_launches++;
            if (quad) e->quad_launches++;
            p.host_flag = nullptr;                   // (the deferred fallback re-uses p: nothing after this launch publishes)
            if (defer) {
                e->spin_used = e->spin_flag != nullptr;
                e->spin_nflags = p.B;
                e->ivf_deferred = p;
                e->ivf_has_deferred = !inline_exact;          // (nothing left to redo when the flagged blocks did it themselves)
                return RII_OK;
            }
            if (inline_exact) continue;
        } else if (!big_all) {
            ScopedTimer t(e, "ivf_coarse", st);
            HIP_TRY(launch_ivf_coarse(p, st));
        }
        if (fused && ivf_exact_lds_supported(e->M, e->Ks, (int) nlist, L)) {
            // flagged queries: exact std::partial_sort emulation with every working set in LDS
            ScopedTimer t(e, "ivf_exact", st);
            HIP_TRY(launch_ivf_exact_lds(p, st));
        } else if ((fused || big_all) && ivf_exact_big_supported(e->M, e->Ks, w, topk)) {
            // nlist or L past the LDS kernel's limits: sequences in global scratch, heaps in LDS -- for the flagged queries of the
            // fused kernel, or for every query when the fused kernel does not cover the shape (its flag array is then NULL)
            int grid = 0;
            RII_TRY(ensure_big_scratch(e, (int) nlist, L, p.B, &grid));
            ScopedTimer t(e, "ivf_exact", st);
            HIP_TRY(launch_ivf_exact_big(p, e->s_big.p, grid, st));
        } else {
            { ScopedTimer t(e, "ivf_plan", st); HIP_TRY(launch_ivf_plan(p, st)); }
            { ScopedTimer t(e, "ivf_scan", st); HIP_TRY(launch_ivf_scan(p, st)); }
            if (topk != 1) { ScopedTimer t(e, "ivf_select", st); HIP_TRY(launch_ivf_select(p, st)); }
        }
    }
    return RII_OK;
}

// the exact-emulation kernels for the queries ivf_fused_kernel flagged (deferred form, see host_query)
int ivf_run_deferred_fallback(rii_engine *e, hipStream_t st)
{
    if (!e->ivf_has_deferred) return RII_OK;
    const IvfParams &p = e->ivf_deferred;
    if (ivf_exact_lds_supported(p.M, p.Ks, p.nlist, p.L)) {
        ScopedTimer t(e, "ivf_exact", st);
        HIP_TRY(launch_ivf_exact_lds(p, st));
    } else if (ivf_exact_big_supported(p.M, p.Ks, p.w, p.topk)) {
        int grid = 0;
        RII_TRY(ensure_big_scratch(e, p.nlist, p.L, p.B, &grid));
        ScopedTimer t(e, "ivf_exact", st);
        HIP_TRY(launch_ivf_exact_big(p, e->s_big.p, grid, st));
    } else {
        { ScopedTimer t(e, "ivf_plan", st); HIP_TRY(launch_ivf_plan(p, st)); }
        { ScopedTimer t(e, "ivf_scan", st); HIP_TRY(launch_ivf_scan(p, st)); }
        if (p.topk != 1) { ScopedTimer t(e, "ivf_select", st); HIP_TRY(launch_ivf_select(p, st)); }
    }
    e->ivf_has_deferred = false;
    return RII_OK;
}

int stage_inputs(rii_engine *e, const float *queries, int64_t B, const int64_t *tids, int64_t S, int topk)
{
    const size_t D = (size_t) e->M * e->Ds;
    RII_TRY(e->s_queries.ensure(std::max<size_t>((size_t) B * D * sizeof(float), 16)));
    RII_TRY(e->s_tids.ensure(std::max<size_t>((size_t) S * sizeof(int64_t), 16)));
    RII_TRY(e->s_out_ids.ensure(std::max<size_t>((size_t) B * topk * sizeof(int64_t), 16)));
    RII_TRY(e->s_out_dists.ensure(std::max<size_t>((size_t) B * topk * sizeof(float), 16)));
    RII_TRY(e->s_out_counts.ensure(std::max<size_t>((size_t) B * sizeof(int64_t), 16)));
    if (B) HIP_TRY(hipMemcpyAsync(e->s_queries.p, queries, (size_t) B * D * sizeof(float), hipMemcpyHostToDevice, e->stream));
    if (S) HIP_TRY(hipMemcpyAsync(e->s_tids.p, tids, (size_t) S * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
    return RII_OK;
}

int check_tids_host(const rii_engine *e, const int64_t *tids, int64_t S)
{
    // The reference documents sorted, duplicate-free ids (docs tutorial.rst:190-204) but never checks: QueryLinear scores
    // the ids in the order given, duplicates included (src/rii.h:218-228), and so does the engine.  QueryIvf's
    // std::binary_search (rii.h:294) is only meaningful on sorted ids; there the engine treats the ids as a set, which
    // is the reference's answer whenever its precondition holds.  Out-of-range ids (an out-of-bounds read in the
    // reference) are the one thing rejected.
    for (int64_t s = 0; s < S; ++s)
        if (tids[s] < 0 || tids[s] >= e->N)
            return set_err(RII_ERR_INVALID, "target id %lld out of range [0, %lld)", (long long) tids[s], (long long) e->N);
    return RII_OK;
}

// A lane's scratch buffers are shared by all calls that use it, so work enqueued on a different stream than the lane's
// previous call must wait for it.  A call on another stream than the active lane's last one takes the parked lane instead
// (two alternating streams: one lane each, no waits).  User streams are only touched while the caller vouches for them
// (inside a call).
int lane_wait(rii_engine *e, ScratchSet &l, hipStream_t st)
{
    if (l.last_stream_valid && l.last_stream != st) {
        if (!l.order_ev) HIP_TRY(hipEventCreateWithFlags(&l.order_ev, hipEventDisableTiming));
        if (l.last_stream == e->stream) HIP_TRY(hipEventRecord(l.order_ev, e->stream));
        HIP_TRY(hipStreamWaitEvent(st, l.order_ev, 0));
    }
    return RII_OK;
}
int begin_on(rii_engine *e, hipStream_t st)
{
    ScratchSet &act = *e;
    if (e->lanes > 1 && act.last_stream_valid && act.last_stream != st) std::swap(act, e->parked);
    RII_TRY(lane_wait(e, act, st));
    act.last_stream = st;
    act.last_stream_valid = true;
    return RII_OK;
}
// index mutations and whole-engine synchronisation: the engine's stream waits for both lanes' in-flight work
int begin_exclusive(rii_engine *e)
{
    RII_TRY(lane_wait(e, e->parked, e->stream));
    e->parked.last_stream_valid = false;
    RII_TRY(lane_wait(e, *e, e->stream));
    e->last_stream = e->stream;
    e->last_stream_valid = true;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RII_OK;
}
int end_on(rii_engine *e, hipStream_t st)
{
    if (st == e->stream) return RII_OK;
    if (!e->order_ev) HIP_TRY(hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e->order_ev, st));
    return RII_OK;
}

void free_all(rii_engine *e)
{
    DevBuf *bufs[] = {&e->d_codewords, &e->d_cnorm, &e->d_codes, &e->d_centers, &e->d_symtab, &e->d_pl_off,
                      &e->d_pl_ids, &e->d_list_len, &e->d_scan_codes, &e->d_scan_perm, &e->d_fcodes, &e->d_lcodes, &e->d_rcent, &e->d_rlcodes,
                      &e->d_rl_toff, &e->d_cand_peak};
    for (DevBuf *b : bufs) b->release();
    if (e->h_cand_peak) (void) hipHostFree(e->h_cand_peak);
    e->h_cand_peak = nullptr;
    e->release_all();
    e->parked.release_all();
    for (auto &kv : e->timers)
        for (auto &pr : kv.second.pending) { (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second); }
    e->timers.clear();
    for (hipEvent_t ev : e->event_pool) (void) hipEventDestroy(ev);
    e->event_pool.clear();
    if (e->stream) (void) hipStreamDestroy(e->stream);
    e->stream = nullptr;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
RII_API const char *rii_last_error(void) { return g_err.c_str(); }
RII_API const char *rii_version(void) { return "rii_amd 0.1.0 (gfx950)"; }
RII_API int rii_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

RII_API int rii_create(const float *codewords, int M, int Ks, int Ds, int verbose, int simd_arch, int device,
                       rii_engine **out)
{
    if (!out) return set_err(RII_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!codewords || M <= 0 || Ks <= 0 || Ds <= 0) return set_err(RII_ERR_INVALID, "codewords must be a non-empty (M,Ks,Ds) array");
    if (Ks > 256) return set_err(RII_ERR_INVALID, "Ks=%d: only Ks <= 256 is supported (uint8 codes; src/pqkmeans.cpp:15-21)", Ks);
    if (simd_arch < RII_SIMD_SSE || simd_arch > RII_SIMD_AVX512) return set_err(RII_ERR_INVALID, "bad simd_arch %d", simd_arch);
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
        return set_err(RII_ERR_HIP, "no HIP device available (%s): the engine has no CPU fallback",
                       he == hipSuccess ? "device count is 0" : hipGetErrorString(he));
    if (device < 0 || device >= ndev) return set_err(RII_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    rii_engine *e = new rii_engine();
    e->M = M; e->Ks = Ks; e->Ds = Ds; e->verbose = verbose; e->arch = simd_arch; e->device = device;
    e->QT = lut_tile_for(M, Ks);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) e->n_cu = prop.multiProcessorCount;
    e->codewords.assign(codewords, codewords + (size_t) M * Ks * Ds);
    int r = RII_OK;
    do {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { r = set_err(RII_ERR_HIP, "hipStreamCreate failed"); break; }
        if ((r = e->d_codewords.ensure(e->codewords.size() * sizeof(float))) != RII_OK) break;
        if (hipMemcpy(e->d_codewords.p, e->codewords.data(), e->codewords.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            r = set_err(RII_ERR_HIP, "codeword upload failed");
            break;
        }
    } while (0);
    if (r != RII_OK) { free_all(e); delete e; return r; }
    if (verbose) {
        const char *names[] = {"sse", "avx", "avx512"};
        printf("rii_amd: device %d (%d CUs), M=%d Ks=%d Ds=%d, table tile QT=%d, reference SIMD order: %s\n", device,
               e->n_cu, M, Ks, Ds, e->QT, names[simd_arch]);
    }
    *out = e;
    return RII_OK;
}

RII_API void rii_destroy(rii_engine *e)
{
    if (!e) return;
    (void) hipSetDevice(e->device);
    free_all(e);
    delete e;
}

RII_API int rii_add_codes(rii_engine *e, const uint8_t *codes, int64_t n, int update_flag)
{
    if (!e || (n > 0 && !codes) || n < 0) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_exclusive(e));
    if (update_flag && e->centers.empty())
        return set_err(RII_ERR_STATE,
                       "reconfigure() must be called before add(vecs=X, update_posting_lists=True). If this is the "
                       "first addition, please call add_configure(vecs=X)");
    if ((e->N + n) > (int64_t) INT32_MAX) return set_err(RII_ERR_UNSUPPORTED, "more than 2^31-1 codes per engine (posting ids are int, src/rii.h:82)");
    const int64_t N0 = e->N;
    RII_TRY(append_codes(e, codes, n));
    if (e->verbose) {
        printf("%lld new vectors are added.\nTotal number of codes is %lld\n", (long long) n, (long long) e->N);
    }
    if (update_flag) {
        if (e->verbose) printf("Start to update posting lists\n");
        RII_TRY(update_posting_lists(e, N0, n));
    }
    return RII_OK;
}

RII_API int rii_set_coarse_centers(rii_engine *e, const uint8_t *centers, int64_t nlist)
{
    if (!e || !centers || nlist <= 0) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_exclusive(e));
    e->centers.assign(centers, centers + (size_t) nlist * e->M);
    RII_TRY(upload_centers(e));
    e->lists.assign((size_t) nlist, std::vector<int32_t>());
    for (auto &l : e->lists) l.reserve((size_t) (e->N / nlist));
    e->lists_dirty = true;
    return update_posting_lists(e, 0, e->N);
}

// Centres and posting lists as given (the lists half of the pickle set-state, src/main.cpp:39-52), the codes untouched: an index
// whose lists were built elsewhere -- another rank, a cached file (examples/benchmark/run_sift1b.py:73-99), a synthetic partition.
RII_API int rii_set_posting_lists(rii_engine *e, const uint8_t *centers, int64_t nlist, const int64_t *pl_off, const int32_t *pl_ids)
{
    if (!e || !centers || nlist <= 0 || !pl_off || (pl_off[nlist] > 0 && !pl_ids)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_exclusive(e));
    if (pl_off[0] != 0) return set_err(RII_ERR_INVALID, "pl_off[0] must be 0");
    for (int64_t i = 0; i < nlist; ++i)
        if (pl_off[i + 1] < pl_off[i]) return set_err(RII_ERR_INVALID, "pl_off must be non-decreasing");
    for (int64_t j = 0; j < pl_off[nlist]; ++j)
        if (pl_ids[j] < 0 || (int64_t) pl_ids[j] >= e->N) return set_err(RII_ERR_INVALID, "posting id %d out of range [0, %lld)", pl_ids[j], (long long) e->N);
    if (e->Ks < 256)
        for (size_t j = 0; j < (size_t) nlist * e->M; ++j)
            if ((int) centers[j] >= e->Ks) return set_err(RII_ERR_INVALID, "centre code %d is not below Ks = %d", (int) centers[j], e->Ks);
    e->centers.assign(centers, centers + (size_t) nlist * e->M);
    RII_TRY(upload_centers(e));
    e->lists.assign((size_t) nlist, std::vector<int32_t>());
    for (int64_t i = 0; i < nlist; ++i) e->lists[(size_t) i].assign(pl_ids + pl_off[i], pl_ids + pl_off[i + 1]);
    e->lists_dirty = true;
    return RII_OK;
}

RII_API int rii_set_state(rii_engine *e, const uint8_t *centers, int64_t nlist, const uint8_t *codes, int64_t N,
                          const int64_t *pl_off, const int32_t *pl_ids)
{
    if (!e || nlist < 0 || N < 0) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_exclusive(e));
    e->codes.clear(); e->N = 0;
    RII_TRY(append_codes(e, codes, N));
    e->centers.assign(centers, centers + (size_t) nlist * e->M);
    RII_TRY(upload_centers(e));
    e->lists.assign((size_t) nlist, std::vector<int32_t>());
    for (int64_t i = 0; i < nlist; ++i) e->lists[(size_t) i].assign(pl_ids + pl_off[i], pl_ids + pl_off[i + 1]);
    e->lists_dirty = true;
    return RII_OK;
}

// RiiCpp::Reconfigure, src/rii.h:108-156.  Sampling and centre initialisation call the very same libstdc++
// facilities as the reference (std::shuffle with default_random_engine(123) / mt19937(0)) so that the
// permutations are identical; assignment, histogram and vote run on the GPU.
RII_API int rii_reconfigure(rii_engine *e, int nlist, int iter)
{
    if (!e) return set_err(RII_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_exclusive(e));
    if (nlist <= 0 || (int64_t) nlist > e->N)
        return set_err(RII_ERR_INVALID, "reconfigure: need 0 < nlist=%d <= N=%lld (src/rii.h:110-111)", nlist, (long long) e->N);
    if (iter < 0) return set_err(RII_ERR_INVALID, "iter must be >= 0");
    const int M = e->M, Ks = e->Ks;
    const size_t len = (size_t) std::min<int64_t>(e->N, (int64_t) nlist * 100);
    if (e->verbose) printf("The number of vectors used for training of coarse centers: %zu\n", len);
    // (1) sampling, rii.h:113-133
    std::vector<size_t> ids((size_t) e->N);
    std::iota(ids.begin(), ids.end(), 0);
    std::shuffle(ids.begin(), ids.end(), std::default_random_engine(123));
    ids.resize(len);
    std::vector<uint8_t> sample(len * M);
    for (size_t i = 0; i < len; ++i) memcpy(&sample[i * M], &e->codes[ids[i] * M], (size_t) M);
    // (2) PQk-means, pqkmeans.cpp:46-133
    if (e->verbose) printf("Start to run PQk-means\n");
    std::vector<int> pick(len);
    std::iota(pick.begin(), pick.end(), 0);
    std::mt19937 random_engine(0);
    std::shuffle(pick.begin(), pick.end(), random_engine);
    std::vector<uint8_t> centers((size_t) nlist * M);
    for (int k = 0; k < nlist; ++k) memcpy(&centers[(size_t) k * M], &sample[(size_t) pick[(size_t) k] * M], (size_t) M);

    RII_TRY(ensure_symtab(e));
    RII_TRY(e->s_sample.ensure(sample.size()));
    RII_TRY(e->d_centers.ensure(centers.size()));
    RII_TRY(e->s_assign.ensure(len * sizeof(int32_t)));
    RII_TRY(e->s_hist.ensure((size_t) nlist * M * Ks * sizeof(int32_t)));
    RII_TRY(e->s_cnt.ensure((size_t) nlist * sizeof(int32_t)));
    hipStream_t st = e->stream;
    HIP_TRY(hipMemcpyAsync(e->s_sample.p, sample.data(), sample.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(e->d_centers.p, centers.data(), centers.size(), hipMemcpyHostToDevice, st));
    for (int itr = 0; itr < iter; ++itr) {
        if (e->verbose) printf("Iteration start: %d / %d\n", itr, iter);
        auto t0 = std::chrono::system_clock::now();
        {
            ScopedTimer t(e, "assign", st);
            HIP_TRY(launch_assign(e->s_sample.as<uint8_t>(), (int64_t) len, M, Ks, e->d_symtab.as<float>(),
                                  e->d_centers.as<uint8_t>(), nlist, e->s_assign.as<int32_t>(), st));
        }
        if (itr != iter - 1) {
            HIP_TRY(hipMemsetAsync(e->s_hist.p, 0, (size_t) nlist * M * Ks * sizeof(int32_t), st));
            HIP_TRY(hipMemsetAsync(e->s_cnt.p, 0, (size_t) nlist * sizeof(int32_t), st));
            HIP_TRY(launch_pqk_hist(e->s_sample.as<uint8_t>(), e->s_assign.as<int32_t>(), (int64_t) len, M, Ks,
                                    e->s_hist.as<int32_t>(), e->s_cnt.as<int32_t>(), st));
            HIP_TRY(launch_pqk_vote(e->s_hist.as<int32_t>(), e->s_cnt.as<int32_t>(), e->d_symtab.as<float>(), nlist, M, Ks,
                                    e->d_centers.as<uint8_t>(), st));
        }
        if (e->verbose) {
            HIP_TRY(hipStreamSynchronize(st));
            printf("find_nn+update_center_time,%lld\n",
                   (long long) std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - t0).count());
        }
    }
    // (3) centres back to the host mirror, rii.h:143
    e->centers.resize(centers.size());
    HIP_TRY(hipMemcpyAsync(e->centers.data(), e->d_centers.p, centers.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // (4) posting lists, rii.h:147-155
    if (e->verbose) printf("Start to update posting lists\n");
    e->lists.assign((size_t) nlist, std::vector<int32_t>());
    for (auto &l : e->lists) l.reserve((size_t) (e->N / nlist));
    e->lists_dirty = true;
    return update_posting_lists(e, 0, e->N);
}

RII_API int rii_clear(rii_engine *e)
{
    if (!e) return set_err(RII_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> guard(e->mu);
    e->centers.clear();
    e->codes.clear();
    e->lists.clear();
    e->N = 0;
    e->lists_dirty = true;
    return RII_OK;
}

RII_API int64_t rii_get_N(const rii_engine *e) { return e ? e->N : 0; }
RII_API int64_t rii_get_nlist(const rii_engine *e) { return e ? nlist_of(e) : 0; }
RII_API int rii_get_M(const rii_engine *e) { return e ? e->M : 0; }
RII_API int rii_get_Ks(const rii_engine *e) { return e ? e->Ks : 0; }
RII_API int rii_get_Ds(const rii_engine *e) { return e ? e->Ds : 0; }
RII_API int rii_get_verbose(const rii_engine *e) { return e ? e->verbose : 0; }
RII_API int rii_set_verbose(rii_engine *e, int verbose)
{
    if (!e) return set_err(RII_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> guard(e->mu);
    e->verbose = verbose;
    return RII_OK;
}
RII_API int rii_get_codewords(const rii_engine *e, float *out)
{
    if (!e || !out) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    memcpy(out, e->codewords.data(), e->codewords.size() * sizeof(float));
    return RII_OK;
}
RII_API int rii_get_codes(const rii_engine *e, uint8_t *out)
{
    if (!e || (!out && e->N)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    if (e->N) memcpy(out, e->codes.data(), e->codes.size());
    return RII_OK;
}
RII_API int rii_get_coarse_centers(const rii_engine *e, uint8_t *out)
{
    if (!e || (!out && !e->centers.empty())) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    if (!e->centers.empty()) memcpy(out, e->centers.data(), e->centers.size());
    return RII_OK;
}
RII_API int rii_get_posting_lists(const rii_engine *e, int64_t *off, int32_t *ids)
{
    if (!e || !off) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    off[0] = 0;
    for (size_t i = 0; i < e->lists.size(); ++i) {
        if (ids && !e->lists[i].empty()) memcpy(ids + off[i], e->lists[i].data(), e->lists[i].size() * sizeof(int32_t));
        off[i + 1] = off[i] + (int64_t) e->lists[i].size();
    }
    return RII_OK;
}

// Host-pointer calls.  Small batches (the reference's one-query-per-call pattern above all) go through ONE pinned
// staging buffer: queries in with a single async H2D, [ids | dists | counts] back with a single D2H -- the pageable
// copies of the generic path cost more than the kernels at that size.
namespace {
constexpr size_t kPinLimit = 8 << 20;
// head of a lane's pinned block: [0, 4096) the kPinFlagWords sequence flags of host_spin -- words that never hold anything but
// sequence numbers -- and [4096, 8192) the tie words of slice_topk_kernel (their own page: ADVICE r3); data follows
constexpr size_t kPinTieOffset = kPinFlagWords * sizeof(unsigned int);
constexpr size_t kPinFlagBytes = 2 * kPinTieOffset;
constexpr size_t kSpinMaxInput = 8192;

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
// bounded wait for the n sequence flags a kernel raises behind its rows in the pinned block; false: not seen (the caller
// synchronises the stream, which is always correct)
bool spin_wait(const volatile unsigned int *flags, int64_t n, unsigned int seq)
{
    bool seen = false;
    for (int spins = 0; spins < 400000 && !seen; ++spins) {
        seen = true;
        for (int64_t b = 0; b < n; ++b) seen = seen && (flags[b] == seq);
        if (!seen) cpu_relax();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return seen;
}

int ensure_pin(rii_engine *e, size_t bytes)
{
    if (bytes <= e->h_pin_cap) return RII_OK;
    if (e->h_pin) HIP_TRY(hipHostFree(e->h_pin));
    e->h_pin = nullptr; e->h_pin_cap = 0;
    const size_t cap = std::max<size_t>(bytes, 64 << 10);
    e->d_pin = nullptr;
    if (hipHostMalloc(&e->h_pin, cap, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||      // (coherent: the kernel-written
        hipHostGetDevicePointer(&e->d_pin, e->h_pin, 0) != hipSuccess || !e->d_pin) {                    //  rows + flags of host_spin)
        // no coherent mapped host memory on this system: plain pinned staging, every call copies and synchronises
        (void) hipGetLastError();
        if (e->h_pin) (void) hipHostFree(e->h_pin);
        e->h_pin = nullptr; e->d_pin = nullptr;
        e->host_spin = 0;
        HIP_TRY(hipHostMalloc(&e->h_pin, cap, hipHostMallocDefault));
    }
    memset(e->h_pin, 0, cap);
    e->h_pin_cap = cap;
    return RII_OK;
}

// runs one host-pointer query call; ivf == false: linear (out_counts unused)
int host_query(rii_engine *e, bool ivf, const float *queries, int64_t B, int topk, const int64_t *tids, int64_t S, int64_t L,
               int64_t *out_ids, float *out_dists, int64_t *out_counts)
{
    const size_t D = (size_t) e->M * e->Ds;
    const size_t q_bytes = (size_t) B * D * sizeof(float);
    const size_t ids_bytes = (size_t) B * topk * sizeof(int64_t), d_bytes = (size_t) B * topk * sizeof(float);
    const size_t c_bytes = ivf ? (size_t) B * sizeof(int64_t) : 0;
    const size_t f_bytes = ivf ? (size_t) B * sizeof(int32_t) : 0;    // flags of ivf_fused_kernel (deferred fallback)
    const size_t out_bytes = ids_bytes + c_bytes + f_bytes + d_bytes; // 8-byte fields first: keeps every field aligned
    const bool small = q_bytes + out_bytes <= kPinLimit;
    hipStream_t st = e->stream;
    if (!small) {
        RII_TRY(stage_inputs(e, queries, B, tids, S, topk));
        if (ivf)
            RII_TRY(query_ivf_dev(e, e->s_queries.as<float>(), B, topk, e->s_tids.as<int64_t>(), S, L, e->s_out_ids.as<int64_t>(),
                                  e->s_out_dists.as<float>(), e->s_out_counts.as<int64_t>(), st));
        else
            RII_TRY(query_linear_dev(e, e->s_queries.as<float>(), B, topk, e->s_tids.as<int64_t>(), S,
                                     e->s_out_ids.as<int64_t>(), e->s_out_dists.as<float>(), st));
        HIP_TRY(hipMemcpyAsync(out_ids, e->s_out_ids.p, ids_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_dists, e->s_out_dists.p, d_bytes, hipMemcpyDeviceToHost, st));
        if (ivf) HIP_TRY(hipMemcpyAsync(out_counts, e->s_out_counts.p, c_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return RII_OK;
    }
    // target ids of a small call ride in the same pinned block and the same H2D copy as the queries (a copy from the caller's
    // pageable array is staged and synchronised by the runtime: +15 us on a 40 us call)
    const size_t t_bytes = (size_t) S * sizeof(int64_t);
    const size_t q_pad = (q_bytes + 15) & ~(size_t) 15;
    const bool pack_tids = S > 0 && q_pad + t_bytes + out_bytes <= kPinLimit;
    const size_t in_bytes = pack_tids ? q_pad + t_bytes : q_pad;
    // the first kPinFlagBytes of the pinned block hold ONLY the per-query sequence flags of host_spin (a fixed place: a flag word
    // must never have held anything but sequence numbers, or stale data could equal the awaited one)
    RII_TRY(ensure_pin(e, kPinFlagBytes + in_bytes + out_bytes));
    RII_TRY(e->s_queries.ensure(std::max<size_t>(in_bytes, 16)));
    RII_TRY(e->s_out_pack.ensure(std::max<size_t>(out_bytes, 16)));
    if (!pack_tids) RII_TRY(e->s_tids.ensure(std::max<size_t>(t_bytes, 16)));
    unsigned char *pin = static_cast<unsigned char *>(e->h_pin) + kPinFlagBytes;
    memcpy(pin, queries, q_bytes);
    if (pack_tids) memcpy(pin + q_pad, tids, t_bytes);
    const bool spin_linear = !ivf && e->host_spin && e->d_pin && in_bytes <= kSpinMaxInput && (size_t) B <= kPinFlagWords &&
                             takes_small_topk(e, B, topk, S);
    // no target ids, exact tables: small_topk_kernel reads the query straight from the pinned block (once per block, through LDS)
    // -- no H2D copy in front of the launch either
    // a few queries on an index too large for that kernel: slice_topk_kernel (one launch; exact ties come back as a flag and the
    // call is redone on the general path)
    const bool slice_linear = !ivf && !spin_linear && e->host_spin && e->d_pin && e->slice_topk && e->QT != 0 && e->lut_mode == RII_LUT_EXACT &&
                              in_bytes <= kSpinMaxInput && slice_topk_supported(e->M, e->Ks, e->Ds, S ? S : e->N, B, topk);
    // a batch on the general path: rows written straight into the pinned block (see below); host_zero_copy: queries read from it too
    // (only where every kernel of the path reads a query O(1) times: the fused table kernel + the codebook re-rank; the per-entry table
    //  kernels of the other shapes and of the small exhaustive batches would fetch it over PCIe thousands of times)
    const bool zc_path = e->scan_mode == 1 && e->fused_tables && qlut_fused_supported(e->M, e->Ks, e->Ds, e->scan_mx) && topk <= rerank_topk_max_k() &&
                         !(topk == 1 && B < e->fast_min_batch);
    const bool zc = !ivf && !spin_linear && !slice_linear && zc_path && e->host_spin && e->d_pin && B <= kMaxBatch && S == 0 && e->lut_mode == RII_LUT_EXACT &&
                    e->QT != 0 && (e->host_zero_copy == 2 || (e->host_zero_copy == 1 && q_bytes <= (128u << 10)));
    const bool batch_pin = zc;
    // inverted index, one-launch form: the fused kernel fetches the query from the pinned block itself (once per block, into LDS)
    const bool spin_ivf = ivf && e->host_spin && e->d_pin && in_bytes <= kSpinMaxInput && (size_t) B <= kPinFlagWords && B < e->fast_min_batch;
    bool ivf_in_place = false;
    if (spin_ivf && S == 0 && e->lut_mode == RII_LUT_EXACT && e->ivf_fused && e->Ds == 4 && e->Ks == 256 && nlist_of(e) > 0) {
        const int64_t nl = nlist_of(e);
        int64_t w = (int64_t) (size_t) std::round((double) L * (double) nl / (double) e->N) + 3;
        if (nl < w) w = nl;
        ivf_in_place = ivf_fused_supported(e->M, e->Ks, (int) nl, w, topk);
    }
    const bool q_in_place = ((spin_linear || slice_linear) && S == 0 && e->lut_mode == RII_LUT_EXACT) || zc || ivf_in_place;
    if (!q_in_place) HIP_TRY(hipMemcpyAsync(e->s_queries.p, pin, pack_tids ? in_bytes : q_bytes, hipMemcpyHostToDevice, st));
    if (S && !pack_tids) HIP_TRY(hipMemcpyAsync(e->s_tids.p, tids, t_bytes, hipMemcpyHostToDevice, st));
    const int64_t *d_tids_in = pack_tids ? reinterpret_cast<const int64_t *>(e->s_queries.as<unsigned char>() + q_pad) : e->s_tids.as<int64_t>();
    unsigned char *dp = e->s_out_pack.as<unsigned char>();
    int64_t *d_ids = reinterpret_cast<int64_t *>(dp);
    int64_t *d_cnt = reinterpret_cast<int64_t *>(dp + ids_bytes);
    int32_t *d_flag = reinterpret_cast<int32_t *>(dp + ids_bytes + c_bytes);
    float *d_d = reinterpret_cast<float *>(dp + ids_bytes + c_bytes + f_bytes);
    unsigned char *pout = pin + in_bytes;
    // (inputs above a few KB are copied by a DMA engine instead of a blit kernel; behind such a copy the flag arrives later than the
    //  stream synchronisation returns -- measured: 3000 target ids 38.6 us with the synchronisation, 41.4 us with the flag)
    if (slice_linear) {
        const int64_t n_codes = S ? S : e->N;
        unsigned char *dp_host = static_cast<unsigned char *>(e->d_pin) + kPinFlagBytes;
        volatile unsigned int *flags = reinterpret_cast<volatile unsigned int *>(e->h_pin);
        volatile int32_t *ties = reinterpret_cast<volatile int32_t *>(static_cast<unsigned char *>(e->h_pin) + kPinTieOffset);
        const unsigned int seq = ++e->spin_seq ? e->spin_seq : ++e->spin_seq;
        RII_TRY(e->s_keys_b.ensure(std::max<size_t>(slice_topk_scratch(n_codes, B, topk), 16)));
        RII_TRY(ensure_small_done(e, st));
        {
            ScopedTimer t(e, "scan", st);
            HIP_TRY(launch_slice_topk(e->d_codes.as<uint8_t>(), n_codes, e->M, e->Ks,
                                      q_in_place ? reinterpret_cast<const float *>(dp_host) : e->s_queries.as<float>(),
                                      e->d_codewords.as<float>(), e->Ds, e->arch, B, topk, S ? d_tids_in : nullptr,
                                      e->s_keys_b.as<unsigned long long>(), e->s_small_done.as<unsigned int>(),
                                      reinterpret_cast<int64_t *>(dp_host + in_bytes), reinterpret_cast<float *>(dp_host + in_bytes + ids_bytes),
                                      reinterpret_cast<int32_t *>(static_cast<unsigned char *>(e->d_pin) + kPinTieOffset), st,
                                      reinterpret_cast<unsigned int *>(e->d_pin), seq));
        }
        const bool seen = spin_wait(flags, B, seq);
        if (!seen) HIP_TRY(hipStreamSynchronize(st));
        bool any_tie = false;
        for (int64_t b = 0; b < B; ++b) any_tie |= (ties[b] != 0);
        if (any_tie) {
            // two of the k + 1 smallest distances of some query are bit-equal: std::partial_sort's order is decided by the general
            // path (tieorder.hip) -- the whole (small) call again
            if (q_in_place) HIP_TRY(hipMemcpyAsync(e->s_queries.p, pin, q_bytes, hipMemcpyHostToDevice, st));
            RII_TRY(query_linear_dev(e, e->s_queries.as<float>(), B, topk, d_tids_in, S, d_ids, d_d, st));
            HIP_TRY(hipMemcpyAsync(pout, dp, out_bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        memcpy(out_ids, pout, ids_bytes);
        memcpy(out_dists, pout + ids_bytes, d_bytes);
        return RII_OK;
    }
    if (spin_linear) {
        // the kernel writes the rows into the pinned block itself and raises one flag per query behind them; the host spins on the
        // flags (bounded: after some tens of milliseconds it falls back to the stream synchronisation, which is always correct)
        unsigned char *dp_host = static_cast<unsigned char *>(e->d_pin) + kPinFlagBytes;
        volatile unsigned int *flags = reinterpret_cast<volatile unsigned int *>(e->h_pin);
        const unsigned int seq = ++e->spin_seq ? e->spin_seq : ++e->spin_seq;          // never 0
        e->spin_flag = reinterpret_cast<unsigned int *>(e->d_pin);
        e->spin_used = false;
        const int r = query_linear_dev(e, q_in_place ? reinterpret_cast<const float *>(dp_host) : e->s_queries.as<float>(), B, topk, d_tids_in, S,
                                       reinterpret_cast<int64_t *>(dp_host + in_bytes),
                                       reinterpret_cast<float *>(dp_host + in_bytes + ids_bytes), st);
        e->spin_flag = nullptr;
        if (r != RII_OK) return r;
        const bool seen = e->spin_used && spin_wait(flags, e->spin_nflags, seq);
        if (!seen) HIP_TRY(hipStreamSynchronize(st));
        memcpy(out_ids, pout, ids_bytes);
        memcpy(out_dists, pout + ids_bytes, d_bytes);
        return RII_OK;
    }
    if (spin_ivf) {
        // the same for the inverted index: every output field (rows, counts, fallback flags) lives in the pinned block; the fused
        // kernel raises a query's sequence flag at each of its exits.  A call that takes another path (spin_used stays false) is
        // simply synchronised -- its kernels wrote the pinned block too.
        unsigned char *dp_host = static_cast<unsigned char *>(e->d_pin) + kPinFlagBytes + in_bytes;
        volatile unsigned int *flags = reinterpret_cast<volatile unsigned int *>(e->h_pin);
        const unsigned int seq = ++e->spin_seq ? e->spin_seq : ++e->spin_seq;
        e->spin_flag = reinterpret_cast<unsigned int *>(e->d_pin);
        e->spin_used = false;
        e->ivf_q_host = ivf_in_place;
        const int r = query_ivf_dev(e, ivf_in_place ? reinterpret_cast<const float *>(static_cast<unsigned char *>(e->d_pin) + kPinFlagBytes) : e->s_queries.as<float>(),
                                    B, topk, d_tids_in, S, L, reinterpret_cast<int64_t *>(dp_host),
                                    reinterpret_cast<float *>(dp_host + ids_bytes + c_bytes + f_bytes),
                                    reinterpret_cast<int64_t *>(dp_host + ids_bytes), st,
                                    reinterpret_cast<int32_t *>(dp_host + ids_bytes + c_bytes));
        e->ivf_q_host = false;
        e->spin_flag = nullptr;
        if (r != RII_OK) return r;
        const bool seen = e->spin_used && spin_wait(flags, e->spin_nflags, seq);
        if (!seen) HIP_TRY(hipStreamSynchronize(st));
        if (e->ivf_has_deferred) {
            const int32_t *fl = reinterpret_cast<const int32_t *>(pout + ids_bytes + c_bytes);
            bool any = false;
            for (int64_t b = 0; b < B; ++b) any |= (fl[b] != 0);
            if (any) {
                RII_TRY(ivf_run_deferred_fallback(e, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            e->ivf_has_deferred = false;
        }
        memcpy(out_ids, pout, ids_bytes);
        memcpy(out_counts, pout + ids_bytes, c_bytes);
        memcpy(out_dists, pout + ids_bytes + c_bytes + f_bytes, d_bytes);
        return RII_OK;
    }
    if (batch_pin) {
        // a batch (round 4, host_zero_copy): the kernels read the queries from the pinned block and write the rows straight into it --
        // no H2D copy, no D2H copy; with fused_rerank the last block of every tile raises a flag behind its rows (no stream
        // synchronisation either)
        unsigned char *dp_host = static_cast<unsigned char *>(e->d_pin) + kPinFlagBytes;
        volatile unsigned int *flags = reinterpret_cast<volatile unsigned int *>(e->h_pin);
        const unsigned int seq = ++e->spin_seq ? e->spin_seq : ++e->spin_seq;
        e->spin_flag = reinterpret_cast<unsigned int *>(e->d_pin);
        e->spin_used = false;
        const int r = query_linear_dev(e, reinterpret_cast<const float *>(dp_host), B, topk, d_tids_in, S,
                                       reinterpret_cast<int64_t *>(dp_host + in_bytes), reinterpret_cast<float *>(dp_host + in_bytes + ids_bytes), st);
        e->spin_flag = nullptr;
        if (r != RII_OK) return r;
        const bool seen = e->spin_used && spin_wait(flags, e->spin_nflags, seq);
        if (!seen) HIP_TRY(hipStreamSynchronize(st));
        memcpy(out_ids, pout, ids_bytes);
        memcpy(out_dists, pout + ids_bytes, d_bytes);
        return RII_OK;
    }
    if (ivf)
        RII_TRY(query_ivf_dev(e, e->s_queries.as<float>(), B, topk, d_tids_in, S, L, d_ids, d_d, d_cnt, st, d_flag));
    else
        RII_TRY(query_linear_dev(e, e->s_queries.as<float>(), B, topk, d_tids_in, S, d_ids, d_d, st));
    HIP_TRY(hipMemcpyAsync(pout, dp, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (ivf && e->ivf_has_deferred) {
        // the fused kernel answered every query unless it raised a flag (exact tie / walk past list w): only then are
        // the emulation kernels launched -- three launches saved on the common path
        const int32_t *flags = reinterpret_cast<const int32_t *>(pout + ids_bytes + c_bytes);
        bool any = false;
        for (int64_t b = 0; b < B; ++b) any |= (flags[b] != 0);
        if (any) {
            RII_TRY(ivf_run_deferred_fallback(e, st));
            HIP_TRY(hipMemcpyAsync(pout, dp, out_bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        e->ivf_has_deferred = false;
    }
    memcpy(out_ids, pout, ids_bytes);
    if (ivf) memcpy(out_counts, pout + ids_bytes, c_bytes);
    memcpy(out_dists, pout + ids_bytes + c_bytes + f_bytes, d_bytes);
    return RII_OK;
}
}  // namespace

RII_API int rii_query_linear(rii_engine *e, const float *queries, int64_t B, int topk, const int64_t *tids, int64_t S,
                             int64_t *out_ids, float *out_dists)
{
    if (!e || (B > 0 && (!queries || !out_ids || !out_dists)) || (S > 0 && !tids)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_on(e, e->stream));
    RII_TRY(check_query_args(e, B, topk, S));
    RII_TRY(check_tids_host(e, tids, S));
    if (B == 0) return RII_OK;
    return host_query(e, false, queries, B, topk, tids, S, 0, out_ids, out_dists, nullptr);
}

static int check_ivf_args(const rii_engine *e, int topk, int64_t L)
{
    if (nlist_of(e) == 0) return set_err(RII_ERR_STATE, "query_ivf needs posting lists: call reconfigure() first");
    if ((int64_t) topk > L || L > e->N)
        return set_err(RII_ERR_INVALID, "need topk <= L <= N: topk=%d, L=%lld, N=%lld (src/rii.h:253)", topk, (long long) L, (long long) e->N);
    return RII_OK;
}

RII_API int rii_query_ivf(rii_engine *e, const float *queries, int64_t B, int topk, const int64_t *tids, int64_t S,
                          int64_t L, int64_t *out_ids, float *out_dists, int64_t *out_counts)
{
    if (!e || (B > 0 && (!queries || !out_ids || !out_dists || !out_counts)) || (S > 0 && !tids)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_on(e, e->stream));
    RII_TRY(check_query_args(e, B, topk, S));
    RII_TRY(check_ivf_args(e, topk, L));
    RII_TRY(check_tids_host(e, tids, S));
    if (B == 0) return RII_OK;
    return host_query(e, true, queries, B, topk, tids, S, L, out_ids, out_dists, out_counts);
}

RII_API int rii_query_linear_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                                 int64_t S, int64_t *d_out_ids, float *d_out_dists, void *stream)
{
    if (!e) return set_err(RII_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(check_query_args(e, B, topk, S));
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    RII_TRY(begin_on(e, st));
    // the ordering event is recorded on the error path too: work already enqueued on `st` still uses the shared scratch
    const int r = query_linear_dev(e, d_queries, B, topk, d_tids, S, d_out_ids, d_out_dists, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

RII_API int rii_query_ivf_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                              int64_t S, int64_t L, int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts,
                              void *stream)
{
    if (!e) return set_err(RII_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(check_query_args(e, B, topk, S));
    RII_TRY(check_ivf_args(e, topk, L));
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    RII_TRY(begin_on(e, st));
    const int r = query_ivf_dev(e, d_queries, B, topk, d_tids, S, L, d_out_ids, d_out_dists, d_out_counts, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

// Queries resident in HBM, rows delivered to the HOST: what SURVEY 8d's metric times ("one batched call incl. table build, scan,
// top-k, device->host of results").  The kernels write the rows straight into the lane's pinned, coherent block; where the last
// kernel of the step raises sequence flags behind them (the fused re-rank of the linear scan: one flag per tile) the call
// returns as soon as the host has seen the flags -- no D2H copy, no stream synchronisation -- otherwise after one
// hipStreamSynchronize.  Synchronous: the rows are in out_* when the call returns (src/main.cpp:17-27: results are
// Python-visible when the call returns).
namespace {
int dev_to_host(rii_engine *e, bool ivf, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S, int64_t L,
                int64_t *out_ids, float *out_dists, int64_t *out_counts, hipStream_t st)
{
    const size_t ids_bytes = (size_t) B * topk * sizeof(int64_t), d_bytes = (size_t) B * topk * sizeof(float);
    const size_t c_bytes = ivf ? (size_t) B * sizeof(int64_t) : 0;
    const size_t out_bytes = ids_bytes + c_bytes + d_bytes;
    if (out_bytes > ((size_t) 64 << 20) || B > kMaxBatch) {         // huge results: device buffers, D2H copies, one synchronisation
        RII_TRY(e->s_out_ids.ensure(std::max<size_t>(ids_bytes, 16)));
        RII_TRY(e->s_out_dists.ensure(std::max<size_t>(d_bytes, 16)));
        RII_TRY(e->s_out_counts.ensure(std::max<size_t>(c_bytes, 16)));
        if (ivf) RII_TRY(query_ivf_dev(e, d_queries, B, topk, d_tids, S, L, e->s_out_ids.as<int64_t>(), e->s_out_dists.as<float>(), e->s_out_counts.as<int64_t>(), st));
        else RII_TRY(query_linear_dev(e, d_queries, B, topk, d_tids, S, e->s_out_ids.as<int64_t>(), e->s_out_dists.as<float>(), st));
        HIP_TRY(hipMemcpyAsync(out_ids, e->s_out_ids.p, ids_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_dists, e->s_out_dists.p, d_bytes, hipMemcpyDeviceToHost, st));
        if (ivf) HIP_TRY(hipMemcpyAsync(out_counts, e->s_out_counts.p, c_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return RII_OK;
    }
    RII_TRY(ensure_pin(e, kPinFlagBytes + out_bytes));
    unsigned char *hp = static_cast<unsigned char *>(e->h_pin) + kPinFlagBytes;
    if (!e->d_pin) {                                                 // no mapped host memory on this system: copy + synchronise
        RII_TRY(e->s_out_pack.ensure(std::max<size_t>(out_bytes, 16)));
        unsigned char *dp = e->s_out_pack.as<unsigned char>();
        if (ivf) RII_TRY(query_ivf_dev(e, d_queries, B, topk, d_tids, S, L, reinterpret_cast<int64_t *>(dp), reinterpret_cast<float *>(dp + ids_bytes + c_bytes),
                                       reinterpret_cast<int64_t *>(dp + ids_bytes), st));
        else RII_TRY(query_linear_dev(e, d_queries, B, topk, d_tids, S, reinterpret_cast<int64_t *>(dp), reinterpret_cast<float *>(dp + ids_bytes), st));
        HIP_TRY(hipMemcpyAsync(hp, dp, out_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        unsigned char *dp = static_cast<unsigned char *>(e->d_pin) + kPinFlagBytes;
        volatile unsigned int *flags = reinterpret_cast<volatile unsigned int *>(e->h_pin);
        const unsigned int seq = ++e->spin_seq ? e->spin_seq : ++e->spin_seq;
        e->spin_used = false;
        int r;
        if (ivf) {
            r = query_ivf_dev(e, d_queries, B, topk, d_tids, S, L, reinterpret_cast<int64_t *>(dp), reinterpret_cast<float *>(dp + ids_bytes + c_bytes),
                              reinterpret_cast<int64_t *>(dp + ids_bytes), st);
        } else {
            e->spin_flag = e->host_spin ? reinterpret_cast<unsigned int *>(e->d_pin) : nullptr;
            r = query_linear_dev(e, d_queries, B, topk, d_tids, S, reinterpret_cast<int64_t *>(dp), reinterpret_cast<float *>(dp + ids_bytes), st);
            e->spin_flag = nullptr;
        }
        if (r != RII_OK) return r;
        const bool seen = e->spin_used && spin_wait(flags, e->spin_nflags, seq);
        if (!seen) HIP_TRY(hipStreamSynchronize(st));
    }
    memcpy(out_ids, hp, ids_bytes);
    if (ivf) memcpy(out_counts, hp + ids_bytes, c_bytes);
    memcpy(out_dists, hp + ids_bytes + c_bytes, d_bytes);
    return RII_OK;
}
}  // namespace

RII_API int rii_query_linear_dev_to_host(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S,
                                         int64_t *out_ids, float *out_dists, void *stream)
{
    if (!e || (B > 0 && (!d_queries || !out_ids || !out_dists)) || (S > 0 && !d_tids)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(check_query_args(e, B, topk, S));
    if (B == 0) return RII_OK;
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    RII_TRY(begin_on(e, st));
    const int r = dev_to_host(e, false, d_queries, B, topk, d_tids, S, 0, out_ids, out_dists, nullptr, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

RII_API int rii_query_ivf_dev_to_host(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S,
                                      int64_t L, int64_t *out_ids, float *out_dists, int64_t *out_counts, void *stream)
{
    if (!e || (B > 0 && (!d_queries || !out_ids || !out_dists || !out_counts)) || (S > 0 && !d_tids)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(check_query_args(e, B, topk, S));
    RII_TRY(check_ivf_args(e, topk, L));
    if (B == 0) return RII_OK;
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    RII_TRY(begin_on(e, st));
    const int r = dev_to_host(e, true, d_queries, B, topk, d_tids, S, L, out_ids, out_dists, out_counts, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

// Database-sharded inverted index (not in the reference; kernel and protocol: ivfshard.hip).
RII_API int rii_ivf_list_lengths_dev(rii_engine *e, const int64_t *d_tids, int64_t S, int64_t S_global, int32_t *d_out_len,
                                     void *stream)
{
    if (!e || !d_out_len || (S > 0 && !d_tids) || S < 0) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    if (nlist_of(e) == 0) return set_err(RII_ERR_STATE, "no posting lists: call reconfigure() / set_coarse_centers() first");
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    RII_TRY(begin_on(e, st));
    // from here on the lane's ordering event is recorded on every path (work already enqueued on `st` uses the shared scratch)
    const int64_t nlist = nlist_of(e);
    int r = sync_lists(e);
    const int32_t *src = e->d_list_len.as<int32_t>();
    if (r == RII_OK && S_global) {   // a target set exists: this rank's share of it may be empty (then every list is empty here)
        r = filter_lists_by_targets(e, d_tids, S, st);
        src = e->s_flen.as<int32_t>();
    }
    if (r == RII_OK && hipMemcpyAsync(d_out_len, src, (size_t) nlist * sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
        r = set_err(RII_ERR_HIP, "copy of the list lengths failed");
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

namespace {
// the bodies of rii_ivf_list_lengths_dev / rii_query_ivf_shard_dev: the caller holds e->mu and has begun on `st`
// d_out_len == NULL (round 6): no copy -- *where = the engine's own (filtered) lengths, for a caller that hands them straight to a
// collective on the same stream
int ivf_list_lengths_locked(rii_engine *e, const int64_t *d_tids, int64_t S, int64_t S_global, int32_t *d_out_len, hipStream_t st,
                            const int32_t **where = nullptr)
{
    const int64_t nlist = nlist_of(e);
    RII_TRY(sync_lists(e));
    const int32_t *src = e->d_list_len.as<int32_t>();
    if (S_global) {      // a target set exists: this rank's share of it may be empty (then every list is empty here)
        RII_TRY(filter_lists_by_targets(e, d_tids, S, st));
        src = e->s_flen.as<int32_t>();
    }
    if (where) *where = src;
    if (d_out_len) HIP_TRY(hipMemcpyAsync(d_out_len, src, (size_t) nlist * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return RII_OK;
}
int ivf_shard_w(const rii_engine *e, int64_t S_global, int64_t L, int64_t N_global)
{
    // w of src/rii.h:266-277 from the global sizes
    const int64_t nlist = nlist_of(e);
    const double wd = (S_global == 0) ? std::round((double) L * (double) nlist / (double) N_global)
                                      : std::round((double) L * (double) nlist / (double) S_global);
    int64_t w = (int64_t) (size_t) wd + 3;
    if (nlist < w) w = nlist;
    return (int) w;
}
int ivf_shard_check(const rii_engine *e, int64_t B, int topk, int64_t S_global, int64_t L, int64_t N_global, int64_t rows, int64_t *w_out)
{
    const int64_t nlist = nlist_of(e);
    if (nlist == 0) return set_err(RII_ERR_STATE, "no posting lists: call reconfigure() / set_coarse_centers() first");
    // the reference's preconditions, on the GLOBAL sizes (src/rii.h:252-253,271)
    if (topk < 1 || (int64_t) topk > L || L > N_global || (S_global != 0 && ((int64_t) topk > S_global || S_global > N_global)))
        return set_err(RII_ERR_INVALID, "need topk <= L <= N and topk <= len(target_ids) <= N on the whole database "
                                        "(topk=%d, L=%lld, N=%lld, S=%lld)", topk, (long long) L, (long long) N_global, (long long) S_global);
    if (L > (int64_t) INT32_MAX - 1) return set_err(RII_ERR_INVALID, "L=%lld: traversal positions are 32-bit", (long long) L);
    const int64_t w = ivf_shard_w(e, S_global, L, N_global);
    // any L (round 5): `rows` best owned candidates per query while they fit the selection buffer, or every owned candidate (rows >= L)
    if (rows < 1 || rows > (int64_t) INT32_MAX - 1 || !ivf_shard_supported(e->M, e->Ks, (int) nlist, L, w, (int) rows))
        return set_err(RII_ERR_INVALID, "rows=%lld: at most %d selected rows per query at this shape, or rows >= L = %lld (every owned candidate)",
                       (long long) rows, ivf_shard_max_select_rows(e->M, e->Ks, (int) nlist, L, w), (long long) L);
    *w_out = w;
    return RII_OK;
}
int ivf_shard_locked(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids, int64_t S, int64_t S_global, int64_t L,
                     int64_t w, const int32_t *d_glen, int G, int rank, int rows, int64_t *d_out_ids, float *d_out_dists, int32_t *d_out_pos,
                     int32_t *d_out_nloc, int64_t *d_out_counts, hipStream_t st, void *d_rec = nullptr, int64_t id_offset = 0, int32_t *d_zero2 = nullptr)
{
    // d_rec (round 6): the exchange record for the B x rows selected rows, written by the kernel next to the plain outputs (NULL: none)
    const int64_t nlist = nlist_of(e);
    RII_TRY(sync_lists(e));
    const int32_t *pl_ids = e->d_pl_ids.as<int32_t>();
    const int32_t *list_len = e->d_list_len.as<int32_t>();
    if (S_global != 0) {   // this rank's share of the target ids (possibly none: every list is then empty here)
        RII_TRY(filter_lists_by_targets(e, d_tids, S, st));
        pl_ids = e->s_fids.as<int32_t>();
        list_len = e->s_flen.as<int32_t>();
    }
    // unfiltered lists: the posting-order copy of the codes (option ivf_list_codes, shared with the fused kernels) makes a list's
    // candidates one coalesced run; where it does not fit the kernel gathers by id
    const uint8_t *lcodes = nullptr;
    if (S_global == 0 && e->ivf_list_codes) {
        if (sync_list_codes(e, st) == RII_OK) lcodes = e->d_lcodes.as<uint8_t>();
        else (void) hipGetLastError();
    }
    // nlist or L past the LDS limits: coarse order + cumulative counts of a query in global scratch, <= 256 MiB per launch
    const size_t per_q = ivf_shard_scratch_per_query(e->M, e->Ks, (int) nlist, L, w);
    const int64_t step = per_q ? std::max<int64_t>(1, std::min<int64_t>(kMaxBatch, ((int64_t) 256 << 20) / (int64_t) per_q)) : kMaxBatch;
    if (per_q) RII_TRY(e->s_big.ensure(per_q * (size_t) std::min<int64_t>(step, B)));
    for (int64_t b0 = 0; b0 < B; b0 += step) {
        const int64_t cur = std::min<int64_t>(step, B - b0);
        const int64_t D = (int64_t) e->M * e->Ds;
        // round 5: the any-L kernel builds its query's exact table itself (exact mode): no table launch, no table round trip
        bool own_tables = e->lut_mode == RII_LUT_EXACT && ivf_shard_builds_tables(e->M, e->Ks, (int) nlist, L, w, rows);
        // round 6: the batch's coarse phase as a pre-pass (four queries per block, one 16-byte LDS read per centre lookup,
        // shard_coarse_quad_kernel): tables + picks for the walk kernel, which then neither scores a centre nor writes a coarse key
        // (measured, tools/r6_shard_pre_ab.py: it pays where the coarse phase is a large part of the walk kernel -- thousands of lists -- and the
        //  batch fills the chip; at nlist = 1024 or a few dozen queries the extra launch costs more than the shared lookups save)
        const bool pre = own_tables && e->shard_pre && (e->shard_pre > 1 || (cur >= 256 && nlist >= 4096)) &&
                         shard_coarse_supported(e->M, e->Ks, e->Ds, (int) nlist, L, w, rows);
        const unsigned long long *picks = nullptr;
        const int32_t *pick_ok = nullptr;
        if (pre) {
            RII_TRY(e->s_lut.ensure((size_t) cur * e->M * e->Ks * sizeof(float)));
            RII_TRY(e->s_coarse_d.ensure((size_t) cur * kShardPickStride * 8));
            RII_TRY(e->s_coarse_i.ensure((size_t) cur * 4));
            e->lut_valid = false;
            ScopedTimer t(e, "shard_coarse", st);
            HIP_TRY(launch_shard_coarse(d_queries + b0 * D, e->d_codewords.as<float>(), e->d_centers.as<uint8_t>(), e->M, e->Ds, e->arch, (int) nlist, w,
                                        cur, e->s_lut.as<float>(), e->s_coarse_d.as<unsigned long long>(), e->s_coarse_i.as<int32_t>(), st,
                                        e->shard_dbg_stop > 10 ? e->shard_dbg_stop - 10 : 0));
            picks = e->s_coarse_d.as<unsigned long long>(); pick_ok = e->s_coarse_i.as<int32_t>();
            own_tables = false;
            e->shard_pre_launches++;
        } else if (!own_tables) RII_TRY(build_lut(e, d_queries + b0 * D, cur, st, false, 1));
        ShardPack pk;
        if (d_rec) {
            const int64_t n = B * (int64_t) rows;
            unsigned char *r = static_cast<unsigned char *>(d_rec);
            pk.rec_pos = reinterpret_cast<int64_t *>(r) + b0 * (int64_t) rows;
            pk.rec_id = reinterpret_cast<int64_t *>(r + (size_t) n * 8) + b0 * (int64_t) rows;
            pk.rec_d = reinterpret_cast<float *>(r + (size_t) n * 16) + b0 * (int64_t) rows;
            pk.id_offset = id_offset;
        }
        if (b0 == 0) pk.zero2 = d_zero2;                    // (the first launch of the batch clears the merge's flag words)
        ScopedTimer t(e, "ivf_shard", st);
        HIP_TRY(launch_ivf_shard(e->d_codes.as<uint8_t>(), e->M, e->Ks, own_tables ? nullptr : e->s_lut.as<float>(), e->d_centers.as<uint8_t>(), (int) nlist,
                                 e->d_pl_off.as<int64_t>(), pl_ids, list_len, d_glen, G, rank, cur, topk, L, w, rows,
                                 d_out_ids + b0 * (int64_t) rows, d_out_dists + b0 * (int64_t) rows, d_out_pos + b0 * (int64_t) rows,
                                 d_out_nloc + b0, d_out_counts + b0, e->s_big.p, st,
                                 own_tables ? d_queries + b0 * D : nullptr, e->d_codewords.as<float>(), e->Ds, e->arch, lcodes,
                                 (e->shard_dbg_stop & 0xff) | ((e->shard_force_replay ? 1 : 0) << 8), picks, pick_ok, (d_rec || pk.zero2) ? &pk : nullptr));
    }
    return RII_OK;
}
}  // namespace

RII_API int rii_query_ivf_shard_dev(rii_engine *e, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                                    int64_t S, int64_t S_global, int64_t L, int64_t N_global, const int32_t *d_glen, int G,
                                    int rank, int rows, int64_t *d_out_ids, float *d_out_dists, int32_t *d_out_pos,
                                    int32_t *d_out_nloc, int64_t *d_out_counts, void *stream)
{
    if (rows <= 0) rows = topk + 1;
    if (!e || B < 0 || (B > 0 && (!d_queries || !d_out_ids || !d_out_dists || !d_out_pos || !d_out_nloc || !d_out_counts)) ||
        !d_glen || G < 1 || rank < 0 || rank >= G || (S > 0 && !d_tids) || S < 0)
        return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    int64_t w = 0;
    RII_TRY(ivf_shard_check(e, B, topk, S_global, L, N_global, rows, &w));
    if (B == 0) return RII_OK;
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    RII_TRY(begin_on(e, st));
    const int r = ivf_shard_locked(e, d_queries, B, topk, d_tids, S, S_global, L, w, d_glen, G, rank, rows, d_out_ids, d_out_dists, d_out_pos,
                                   d_out_nloc, d_out_counts, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

RII_API int rii_ivf_shard_max_select_rows(const rii_engine *e, int64_t L, int64_t N_global, int64_t S_global)
{
    if (!e || L < 1 || N_global < 1 || S_global < 0 || nlist_of(e) == 0) return -1;
    return ivf_shard_max_select_rows(e->M, e->Ks, (int) nlist_of(e), L, ivf_shard_w(e, S_global, L, N_global));
}
RII_API int64_t rii_ivf_shard_replay_scratch_bytes(int64_t nf, int rows)
{
    return (nf < 0 || rows < 1) ? -1 : (int64_t) shard_replay_scratch(nf, rows);
}
RII_API int rii_ivf_shard_replay_ex_dev(const void *d_gathered, int G, int64_t nf, int rows, int topk, int64_t *d_out_ids,
                                        float *d_out_dists, void *d_scratch, int64_t scratch_bytes, void *stream)
{
    if (!d_gathered || G < 1 || nf < 0 || rows < 1 || topk < 1 || topk > rows || (nf > 0 && (!d_out_ids || !d_out_dists)))
        return set_err(RII_ERR_INVALID, "bad arguments");
    const size_t need = shard_replay_scratch(nf, rows);
    if (need && (!d_scratch || scratch_bytes < (int64_t) need))
        return set_err(RII_ERR_INVALID, "rows=%d: the candidate sequences need %lld bytes of scratch (rii_ivf_shard_replay_scratch_bytes)", rows, (long long) need);
    HIP_TRY(launch_shard_replay(d_gathered, G, nf, rows, topk, d_out_ids, d_out_dists, d_scratch, (hipStream_t) stream));
    return RII_OK;
}
RII_API int rii_ivf_shard_replay_dev(const void *d_gathered, int G, int64_t nf, int rows, int topk, int64_t *d_out_ids,
                                     float *d_out_dists, void *stream)
{
    if (rows >= 1 && nf >= 0 && shard_replay_scratch(nf, rows) != 0)
        return set_err(RII_ERR_INVALID, "rows=%d: sequences past the LDS budget need scratch: call rii_ivf_shard_replay_ex_dev", rows);
    return rii_ivf_shard_replay_ex_dev(d_gathered, G, nf, rows, topk, d_out_ids, d_out_dists, nullptr, 0, stream);
}

// Database-sharded linear search, exact ties (not in the reference; kernels and argument: tieorder.hip).
namespace {
// the body of rii_linear_tie_emit_dev: the caller holds e->mu and has begun on `st`
int linear_tie_emit_locked(rii_engine *e, const float *d_queries, int64_t nf, int topk, const int64_t *d_tids, int64_t S,
                           const float *d_bound, int64_t id_offset, int cap, int64_t *d_out_ids, float *d_out_dists,
                           int32_t *d_out_count, hipStream_t st)
{
    const int64_t n = S ? S : e->N;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(d_out_count, 0, (size_t) nf * sizeof(int32_t), st));
        return RII_OK;
    }
    const int64_t D = (int64_t) e->M * e->Ds;
    // Cost: the emit scans the whole shard once per group of flagged queries with ~9 bytes of scratch per code and query, so
    // the group size is what fits 1 GiB of scratch (64 queries up to 1.8 M codes, ONE query per launch at a 125 M-code shard:
    // a tie-heavy batch on a Deep1B-sized shard costs nf full-shard passes -- exactness first; docs: DESIGN.md section 6).
    // Tables above the LDS budget (round 4): scan_wide_kernel writes the exact distances of the group as key rows (8 more bytes per code
    // and query), and the chunk kernels read those instead of staging a table.
    const bool wide = e->QT == 0;
    const int64_t fq_max = std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t) 1 << 30) / (n * (wide ? 17 : 9) + 1)));
    // work list of a group = [count | 0, 1, 2, ...]: the indices are uploaded once per engine, the count is a 4-byte memset in
    // stream order (no host synchronisation inside the loop)
    if (!e->have_ident) {
        int32_t ident[65];
        ident[0] = 0;
        for (int i = 0; i < 64; ++i) ident[i + 1] = i;
        RII_TRY(e->s_ident.ensure(sizeof(ident)));
        HIP_TRY(hipMemcpyAsync(e->s_ident.p, ident, sizeof(ident), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        e->have_ident = true;
    }
    for (int64_t f0 = 0; f0 < nf; f0 += fq_max) {
        const int cur = (int) std::min<int64_t>(fq_max, nf - f0);
        RII_TRY(build_lut(e, d_queries + f0 * D, cur, st, false, 0));
        RII_TRY(e->s_tie_chunk.ensure(linear_tie_chunked_scratch(n, cur)));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) e->s_ident.p, cur, 1, st));
        ScopedTimer t(e, "tie", st);
        const unsigned long long *keyrow = nullptr;
        if (wide) {
            RII_TRY(e->s_keys_a.ensure((size_t) cur * (size_t) n * sizeof(unsigned long long)));
            HIP_TRY(launch_scan_wide(e->d_codes.as<uint8_t>(), n, e->M, e->Ks, e->s_lut.as<float>(), S ? d_tids : nullptr, 0, cur,
                                     e->s_keys_a.as<unsigned long long>(), st));
            keyrow = e->s_keys_a.as<unsigned long long>();
        }
        HIP_TRY(launch_linear_tie_emit(e->d_codes.as<uint8_t>(), n, e->M, e->Ks, e->s_lut.as<float>(), e->lut_qt, 0,
                                       e->s_ident.as<int32_t>() + 1, e->s_ident.as<int>(), S ? d_tids : nullptr, topk, cur,
                                       e->s_tie_chunk.p, S ? 1 : 0, d_bound ? d_bound + f0 : nullptr, id_offset, cap,
                                       d_out_ids + f0 * cap, d_out_dists + f0 * cap, d_out_count + f0, st, keyrow));
    }
    return RII_OK;
}
}  // namespace

RII_API int rii_linear_tie_emit_dev(rii_engine *e, const float *d_queries, int64_t nf, int topk, const int64_t *d_tids, int64_t S,
                                    const float *d_bound, int64_t id_offset, int cap, int64_t *d_out_ids, float *d_out_dists,
                                    int32_t *d_out_count, void *stream)
{
    if (!e || nf < 0 || topk < 1 || cap < 1 || S < 0 || (S > 0 && !d_tids) || (nf > 0 && (!d_queries || !d_out_ids || !d_out_dists || !d_out_count)))
        return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    if (S > e->N) return set_err(RII_ERR_INVALID, "S=%lld must satisfy S <= N", (long long) S);
    if (e->QT == 0 ? !linear_tie_chunked_topk_ok(topk) : !linear_tie_chunked_supported(e->M, e->Ks, topk))
        return set_err(RII_ERR_UNSUPPORTED, "tie emission: topk=%d / M*Ks=%d not supported", topk, e->M * e->Ks);
    if (nf == 0) return RII_OK;
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    RII_TRY(begin_on(e, st));
    const int r = linear_tie_emit_locked(e, d_queries, nf, topk, d_tids, S, d_bound, id_offset, cap, d_out_ids, d_out_dists, d_out_count, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}
RII_API int64_t rii_linear_tie_record_bytes(int64_t nf, int cap) { return (int64_t) linear_tie_record_bytes(nf, cap); }
RII_API int rii_linear_tie_replay_dev(const void *d_gathered, int G, int64_t nf, int cap, int topk, int64_t *d_out_ids,
                                      float *d_out_dists, void *stream)
{
    if (!d_gathered || G < 1 || nf < 0 || cap < 1 || topk < 1 || topk > 1024 || (int64_t) G * cap >= ((int64_t) 1 << 32) ||
        (nf > 0 && (!d_out_ids || !d_out_dists)))
        return set_err(RII_ERR_INVALID, "bad arguments");
    HIP_TRY(launch_linear_shard_replay(d_gathered, G, nf, cap, topk, d_out_ids, d_out_dists, (hipStream_t) stream));
    return RII_OK;
}

// Database sharding (not in the reference): merge of the all-gathered per-shard top-k rows.  Stateless.
RII_API int64_t rii_merge_record_bytes(int64_t B, int k, int payload) { return (int64_t) merge_record_bytes(B, k, payload); }
RII_API int rii_merge_topk_dev(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, int64_t *d_out_keys,
                               float *d_out_dists, int64_t *d_out_payload, void *stream)
{
    if (!d_gathered || G < 1 || B < 0 || k < 1 || k_out < 1 || k_out > G * k || (B > 0 && (!d_out_keys || !d_out_dists)) ||
        (payload && B > 0 && !d_out_payload))
        return set_err(RII_ERR_INVALID, "bad arguments");
    if ((int64_t) G * k > merge_topk_max_keys())
        return set_err(RII_ERR_UNSUPPORTED, "merge of %d x %d keys per query exceeds %d", G, k, merge_topk_max_keys());
    HIP_TRY(launch_merge_topk(d_gathered, G, B, k, k_out, payload, d_out_keys, d_out_dists, d_out_payload, (hipStream_t) stream));
    return RII_OK;
}

RII_API int rii_merge_topk_ex_dev(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, const int64_t *id_offsets,
                                  int64_t *d_out_keys, float *d_out_dists, int64_t *d_out_payload, int tie_cols,
                                  int32_t *d_out_tie, int32_t *d_out_any, void *stream)
{
    if (!d_gathered || G < 1 || B < 0 || k < 1 || k_out < 1 || k_out > G * k || (B > 0 && (!d_out_keys || !d_out_dists)) ||
        (payload && B > 0 && !d_out_payload) || tie_cols < 0 || tie_cols > k_out || (id_offsets && G > 64))
        return set_err(RII_ERR_INVALID, "bad arguments");
    if ((int64_t) G * k > merge_topk_max_keys())
        return set_err(RII_ERR_UNSUPPORTED, "merge of %d x %d keys per query exceeds %d", G, k, merge_topk_max_keys());
    if (k == 1 && k_out == 1 && !payload && !d_out_tie && !d_out_any && G <= 64) {   // one row per rank: one thread per query (comm.hip)
        HIP_TRY(launch_merge_top1(d_gathered, G, B, id_offsets, d_out_keys, d_out_dists, (hipStream_t) stream));
        return RII_OK;
    }
    HIP_TRY(launch_merge_topk(d_gathered, G, B, k, k_out, payload, d_out_keys, d_out_dists, d_out_payload, (hipStream_t) stream,
                              id_offsets, tie_cols, d_out_tie, d_out_any));
    return RII_OK;
}

// The merge of records that carry the 16-byte header {int64 id offset of the rank's shard, int32 status, pad} in front of the rows
// (round 5): what rii_query_linear_dbsharded_dev / rii_query_ivf_dbsharded_dev run behind their all-gather, for callers that run the
// collective themselves -- any G, any k (more than 8192 rows per query are sorted in d_scratch: rii_merge_hdr_scratch_bytes()), the
// shard offsets read from the headers, a non-zero status anywhere poisons every row (ids -2, distances NaN; bit 1 of *d_out_any).
RII_API int64_t rii_merge_hdr_record_bytes(int64_t B, int k, int payload) { return (B < 0 || k < 1) ? -1 : (int64_t) merge_record_bytes(B, k, payload) + kRecHeader; }
RII_API int64_t rii_merge_hdr_scratch_bytes(int G, int64_t B, int k) { return (G < 1 || B < 0 || k < 1) ? -1 : (int64_t) merge_topk_scratch(G, B, k); }
RII_API int rii_merge_topk_hdr_dev(const void *d_gathered, int G, int64_t B, int k, int k_out, int payload, int64_t *d_out_keys, float *d_out_dists,
                                   int64_t *d_out_payload, int tie_cols, int32_t *d_out_tie, int32_t *d_out_any, void *d_scratch,
                                   int64_t scratch_bytes, void *stream)
{
    if (!d_gathered || G < 1 || B < 0 || k < 1 || k_out < 1 || (int64_t) k_out > (int64_t) G * k || (int64_t) G * k >= ((int64_t) 1 << 31) ||
        (B > 0 && (!d_out_keys || !d_out_dists)) || (payload && B > 0 && !d_out_payload) || tie_cols < 0 || tie_cols > k_out)
        return set_err(RII_ERR_INVALID, "bad arguments");
    const size_t need = merge_topk_scratch(G, B, k);
    if (need && (!d_scratch || scratch_bytes < (int64_t) need))
        return set_err(RII_ERR_INVALID, "%d x %d rows per query need %lld bytes of scratch (rii_merge_hdr_scratch_bytes)", G, k, (long long) need);
    if (k == 1 && k_out == 1 && !payload && !d_out_tie && !d_out_any) {            // one row per rank: one thread per query (comm.hip)
        HIP_TRY(launch_merge_top1(d_gathered, G, B, nullptr, d_out_keys, d_out_dists, (hipStream_t) stream, kRecHeader));
        return RII_OK;
    }
    HIP_TRY(launch_merge_topk(d_gathered, G, B, k, k_out, payload, d_out_keys, d_out_dists, d_out_payload, (hipStream_t) stream, nullptr, tie_cols,
                              d_out_tie, d_out_any, kRecHeader, d_scratch));
    return RII_OK;
}

// Round 6: merge + finish of the database-sharded inverted index's TOP-1 batch in one launch, for callers who gather themselves (what
// rii_query_ivf_dbsharded_dev runs behind its all-gather): records with payload and header, k = 2 rows per query and rank
RII_API int rii_ivf_merge_top1_hdr_dev(const void *d_gathered, int G, int64_t B, const int64_t *d_counts, int64_t *d_out_ids, float *d_out_dists,
                                       int64_t *d_out_counts, int32_t *d_out_any, void *stream)
{
    if (!d_gathered || G < 1 || B < 0 || (B > 0 && (!d_counts || !d_out_ids || !d_out_dists || !d_out_counts))) return set_err(RII_ERR_INVALID, "bad arguments");
    HIP_TRY(launch_ivf_merge_top1(d_gathered, G, B, kRecHeader, d_counts, d_out_ids, d_out_dists, d_out_counts, nullptr, d_out_any, (hipStream_t) stream));
    return RII_OK;
}

// Query sharding, for callers that run the collective themselves: the record of a rank and the unpack of the G gathered records
// (what rii_query_*_qsharded_dev do internally).  Stateless.
RII_API int64_t rii_qshard_begin(int64_t B, int G, int rank) { return (G < 1 || rank < 0 || rank > G || B < 0) ? -1 : qshard_begin(B, G, rank); }
RII_API int64_t rii_qshard_record_bytes(int64_t B, int G, int k, int counts) { return (G < 1 || B < 0 || k < 1) ? -1 : (int64_t) qshard_record_bytes(B, G, k, counts); }
RII_API int rii_qshard_unpack_dev(const void *d_gathered, int64_t B, int G, int k, int counts, int64_t *d_out_ids, float *d_out_dists,
                                  int64_t *d_out_counts, void *stream)
{
    if (!d_gathered || B < 0 || G < 1 || k < 1 || (B > 0 && (!d_out_ids || !d_out_dists)) || (counts && B > 0 && !d_out_counts))
        return set_err(RII_ERR_INVALID, "bad arguments");
    HIP_TRY(launch_qshard_unpack(d_gathered, B, G, k, counts, d_out_ids, d_out_dists, d_out_counts, (hipStream_t) stream));
    return RII_OK;
}

// =====================================================================================================
// Multi-GPU behind the C ABI (round 4; protocol and kernels: comm.hip, merge.hip, tieorder.hip).  One rii_comm per process and
// GPU = one RCCL communicator; the sharded entry points enqueue  engine kernels -> ONE ncclAllGather of pre-sized records ->
// unpack / merge kernel  on the caller's stream.  Not in the reference (it has no multi-device code): the parity
// target is the single-index answer (src/rii.h:195-242, :244-326) on the concatenated database / the whole batch.
// =====================================================================================================
struct rii_comm {
    void *nccl = nullptr;
    int rank = 0, G = 1, device = 0;
    DevBuf rec, gathered, tmp_i, tmp_d, mi, md, tie, anyf, fsel, qf, bound, rec2, gg, r_i, r_d, lens, status_dev, seq;
    // round 5: every record of the database-sharded calls starts with a kRecHeader-byte header {int64 id offset of this rank's shard,
    // int32 status}: the offsets travel with the rows (nothing cached that another index on the same communicator could leave stale,
    // no collective that only some ranks issue), and so does a rank-local failure.  What this rank last wrote into the header of
    // c->rec (a rank-local cache of a rank-local write):
    void *hdr_at = nullptr;
    int64_t hdr_offset = -1;
    int hdr_status = -1;
    bool broken = false;                  // a collective itself failed: the ranks may be out of step, every later call is refused
    std::mutex mu;
};

RII_API int rii_comm_unique_id(void *id_out)
{
    if (!id_out) return set_err(RII_ERR_INVALID, "id_out is NULL");
    if (const char *err = comm_unique_id(id_out)) return set_err(RII_ERR_HIP, "RCCL: %s", err);
    return RII_OK;
}

RII_API int rii_comm_init(const void *id, int rank, int nranks, int device, rii_comm **out)
{
    if (!out) return set_err(RII_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return set_err(RII_ERR_INVALID, "bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(RII_ERR_HIP, "no HIP device available");
    if (device < 0 || device >= ndev) return set_err(RII_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    rii_comm *c = new rii_comm();
    c->rank = rank; c->G = nranks; c->device = device;
    if (const char *err = comm_create(id, rank, nranks, &c->nccl)) {
        delete c;
        return set_err(RII_ERR_HIP, "RCCL communicator: %s", err);
    }
    *out = c;
    return RII_OK;
}

RII_API void rii_comm_destroy(rii_comm *c)
{
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipDeviceSynchronize();
    comm_destroy(c->nccl);
    DevBuf *bufs[] = {&c->rec, &c->gathered, &c->tmp_i, &c->tmp_d, &c->mi, &c->md, &c->tie, &c->anyf, &c->fsel, &c->qf, &c->bound,
                      &c->rec2, &c->gg, &c->r_i, &c->r_d, &c->lens, &c->status_dev, &c->seq};
    for (DevBuf *b : bufs) b->release();
    delete c;
}
RII_API int rii_comm_rank(const rii_comm *c) { return c ? c->rank : -1; }
RII_API int rii_comm_size(const rii_comm *c) { return c ? c->G : 0; }

namespace {
int comm_gather(rii_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st)
{
    if (const char *err = comm_all_gather(c->nccl, d_send, d_recv, bytes, st)) {
        c->broken = true;                      // the ranks may be out of step from here on
        return set_err(RII_ERR_HIP, "ncclAllGather: %s (the communicator is unusable after a failed collective: destroy it)", err);
    }
    return RII_OK;
}
int comm_usable(const rii_comm *c)
{
    if (c->broken) return set_err(RII_ERR_STATE, "this communicator saw a failed collective: the ranks may be out of step; destroy it");
    return RII_OK;
}
// the header of c->rec: written when it changes only (one small host-to-device copy; the common case is "same shard, status 0")
int comm_set_header(rii_comm *c, int64_t id_offset, int status, hipStream_t st)
{
    if (c->hdr_at == c->rec.p && c->hdr_offset == id_offset && c->hdr_status == status) return RII_OK;
    struct { int64_t off; int32_t status, pad; } h = {id_offset, status, 0};
    static_assert(sizeof(h) == kRecHeader, "record header");
    c->hdr_at = nullptr;
    HIP_TRY(hipMemcpyAsync(c->rec.p, &h, sizeof(h), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));         // (`h` lives on this stack frame)
    c->hdr_at = c->rec.p; c->hdr_offset = id_offset; c->hdr_status = status;
    return RII_OK;
}
// Collective agreement for the RARE phases that talk to the host anyway (exact-tie replays, the collect-all route): every rank
// contributes its local return code behind ONE 4-byte all-gather, so a rank-local failure (an allocation that did not fit, ...)
// makes every rank return an error from the same call instead of leaving its peers inside the next all-gather.
int comm_agree(rii_comm *c, int local_rc, hipStream_t st)
{
    const std::string local_msg = g_err;
    int32_t mine = local_rc;
    std::vector<int32_t> all((size_t) c->G, 0);
    if (c->status_dev.ensure((size_t) (c->G + 1) * sizeof(int32_t)) != RII_OK) {
        c->broken = true;                      // cannot even take part: the peers will wait in their all-gather
        return set_err(RII_ERR_HIP, "no memory for the status exchange (the communicator is unusable)");
    }
    int32_t *sd = c->status_dev.as<int32_t>();
    if (hipMemcpyAsync(sd + c->G, &mine, 4, hipMemcpyHostToDevice, st) != hipSuccess) { c->broken = true; return set_err(RII_ERR_HIP, "copy failed"); }
    RII_TRY(comm_gather(c, sd + c->G, sd, 4, st));
    if (hipMemcpyAsync(all.data(), sd, (size_t) c->G * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        c->broken = true;
        return set_err(RII_ERR_HIP, "copy failed");
    }
    if (local_rc != RII_OK) { g_err = local_msg; return local_rc; }
    for (int g = 0; g < c->G; ++g)
        if (all[(size_t) g] != RII_OK) return set_err(RII_ERR_STATE, "rank %d failed this sharded call (code %d); every rank returns, the communicator stays in step", g, (int) all[(size_t) g]);
    return RII_OK;
}
// shared tail of the query-sharded calls: this rank's rows are in c->rec.  A rank whose own engine call failed (local_rc) still takes
// part -- its rows go out as all-ones bytes (ids -1, counts -1, distances NaN), so its peers return from the same call with those
// rows poisoned instead of waiting in the all-gather -- and returns its error afterwards
int qshard_exchange(rii_comm *c, int local_rc, int64_t B, int topk, int counts, int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts, hipStream_t st)
{
    const std::string local_msg = g_err;
    const size_t rec_bytes = qshard_record_bytes(B, c->G, topk, counts);
    if (local_rc != RII_OK && hipMemsetAsync(c->rec.p, 0xff, rec_bytes, st) != hipSuccess) { c->broken = true; return set_err(RII_ERR_HIP, "memset failed"); }
    RII_TRY(comm_gather(c, c->rec.p, c->gathered.p, rec_bytes, st));
    HIP_TRY(launch_qshard_unpack(c->gathered.p, B, c->G, topk, counts, d_out_ids, d_out_dists, d_out_counts, st));
    if (local_rc != RII_OK) { g_err = local_msg; return local_rc; }
    return RII_OK;
}
}  // namespace

// Query sharding: the index is replicated, rank r answers rows [qshard_begin(r), qshard_begin(r + 1)) of the batch -- every rank
// passes the SAME d_queries [B, D] -- and ONE all-gather of the packed rows gives every rank all B rows.
RII_API int rii_query_linear_qsharded_dev(rii_engine *e, rii_comm *c, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                                          int64_t S, int64_t *d_out_ids, float *d_out_dists, void *stream)
{
    if (!e || !c || (B > 0 && (!d_queries || !d_out_ids || !d_out_dists)) || (S > 0 && !d_tids)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> gc(c->mu);
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(comm_usable(c));
    RII_TRY(check_query_args(e, B, topk, S));        // (the index is replicated: the same outcome on every rank)
    if (B == 0) return RII_OK;
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    const int64_t s0 = qshard_begin(B, c->G, c->rank), n = qshard_begin(B, c->G, c->rank + 1) - s0, nmax = (B + c->G - 1) / c->G;
    const size_t rec_bytes = qshard_record_bytes(B, c->G, topk, 0);
    if (c->rec.ensure(rec_bytes) != RII_OK || c->gathered.ensure(rec_bytes * (size_t) c->G) != RII_OK) {
        c->broken = true;
        return set_err(RII_ERR_HIP, "no memory for the exchange records (the communicator is unusable: the peers wait in their all-gather)");
    }
    c->hdr_at = nullptr;                       // (the record of this call starts where the database-sharded calls keep their header)
    RII_TRY(begin_on(e, st));
    int r = RII_OK;
    if (n > 0)
        r = query_linear_dev(e, d_queries + s0 * (int64_t) (e->M * e->Ds), n, topk, d_tids, S, c->rec.as<int64_t>(),
                             reinterpret_cast<float *>(c->rec.as<unsigned char>() + (size_t) nmax * topk * 8), st);
    r = qshard_exchange(c, r, B, topk, 0, d_out_ids, d_out_dists, nullptr, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

RII_API int rii_query_ivf_qsharded_dev(rii_engine *e, rii_comm *c, const float *d_queries, int64_t B, int topk, const int64_t *d_tids,
                                       int64_t S, int64_t L, int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts, void *stream)
{
    if (!e || !c || (B > 0 && (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts)) || (S > 0 && !d_tids)) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> gc(c->mu);
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(comm_usable(c));
    RII_TRY(check_query_args(e, B, topk, S));
    RII_TRY(check_ivf_args(e, topk, L));
    if (B == 0) return RII_OK;
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    const int64_t s0 = qshard_begin(B, c->G, c->rank), n = qshard_begin(B, c->G, c->rank + 1) - s0, nmax = (B + c->G - 1) / c->G;
    const size_t rec_bytes = qshard_record_bytes(B, c->G, topk, 1);
    if (c->rec.ensure(rec_bytes) != RII_OK || c->gathered.ensure(rec_bytes * (size_t) c->G) != RII_OK) {
        c->broken = true;
        return set_err(RII_ERR_HIP, "no memory for the exchange records (the communicator is unusable: the peers wait in their all-gather)");
    }
    c->hdr_at = nullptr;
    RII_TRY(begin_on(e, st));
    int r = RII_OK;
    unsigned char *rp = c->rec.as<unsigned char>();
    if (n > 0)
        r = query_ivf_dev(e, d_queries + s0 * (int64_t) (e->M * e->Ds), n, topk, d_tids, S, L, reinterpret_cast<int64_t *>(rp),
                          reinterpret_cast<float *>(rp + (size_t) nmax * topk * 8 + (size_t) nmax * 8),
                          reinterpret_cast<int64_t *>(rp + (size_t) nmax * topk * 8), st);
    r = qshard_exchange(c, r, B, topk, 1, d_out_ids, d_out_dists, d_out_counts, st);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

// Database sharding, linear search: rank r's engine holds the codes with global ids [id_offset, id_offset + N_local) (contiguous id
// ranges in rank order); every rank answers the WHOLE batch on its shard (k + 1 rows per query; one row for top-1: a heap of one
// keeps the first minimum in index order = the smallest id), ONE all-gather, and every rank merges the G records under (distance,
// global id).  Where two of the merged k + 1 best distances are bit-equal the reference's order is std::partial_sort's over ALL
// distances in index order: those queries (d_out_tie, identical on every rank) are replayed exactly -- every rank emits, in index
// order, the codes that can touch the heap (tie_cap rows per query and rank), the lists are all-gathered in rank order and one
// wave per query replays the library's heap (tieorder.hip).  Top-1 is asynchronous on the stream; top-k reads ONE word per batch
// on the host (is any query flagged?) and is synchronous only in that.  d_tids_local: this rank's share of the target ids as
// LOCAL ids (S_global = size of the whole target set, 0 = none; a rank may own none: S_local == 0).
//
// Collectives and failures (round 5): every decision that leads to a collective is taken from arguments that are the same
// on every rank; everything rank-local (this shard's size, allocations, the engine's own call) is folded into `lr` and the rank STILL
// takes part in the batch's all-gather, with a non-zero status in its record header -- the merge kernels of every rank then poison
// the batch (ids -2, distances NaN) and the top-k path returns RII_ERR_STATE everywhere; the rare replay phase agrees on a status
// word first (comm_agree).  The ranks stay in step; only a failed collective itself marks the communicator unusable.
RII_API int rii_query_linear_dbsharded_dev(rii_engine *e, rii_comm *c, int64_t id_offset, const float *d_queries, int64_t B, int topk,
                                           const int64_t *d_tids_local, int64_t S_local, int64_t S_global, int64_t *d_out_ids,
                                           float *d_out_dists, int32_t *d_out_tie, int32_t *d_out_overflow, int tie_cap, void *stream)
{
    if (!e || !c || B < 0 || topk < 1 || S_local < 0 || S_global < 0 || (S_local > 0 && !d_tids_local) || S_local > S_global + (S_global == 0 ? S_local : 0) ||
        (B > 0 && (!d_queries || !d_out_ids || !d_out_dists)) || id_offset < 0)
        return set_err(RII_ERR_INVALID, "bad arguments");
    if (S_global == 0 && S_local != 0) return set_err(RII_ERR_INVALID, "S_local=%lld target ids but S_global=0", (long long) S_local);
    std::lock_guard<std::mutex> gc(c->mu);
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(comm_usable(c));
    if (B == 0) return RII_OK;
    const int G = c->G;
    const int rows = topk == 1 ? 1 : topk + 1;
    if ((int64_t) G * rows >= ((int64_t) 1 << 31)) return set_err(RII_ERR_INVALID, "G * (topk + 1) = %lld rows per query", (long long) G * rows);
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    const size_t rec_bytes = merge_record_bytes(B, rows, 0), stride = rec_bytes + kRecHeader;
    // without its own record and the gathered ones a rank cannot take part at all
    if (c->rec.ensure(stride) != RII_OK || c->gathered.ensure(stride * (size_t) G) != RII_OK) {
        c->broken = true;
        return set_err(RII_ERR_HIP, "no memory for the exchange records (the communicator is unusable: the peers wait in their all-gather)");
    }
    // ---- rank-local from here: failures go into lr, the collective below is issued regardless ----
    int lr = RII_OK;
    if (S_local > e->N) lr = set_err(RII_ERR_INVALID, "S_local=%lld must satisfy S <= N", (long long) S_local);
    const int64_t n_local = S_global ? S_local : e->N;
    const int k_local = (int) std::min<int64_t>(rows, n_local);
    const size_t msc = merge_topk_scratch(G, B, rows);
    if (lr == RII_OK && topk > 1 &&
        ((lr = c->mi.ensure((size_t) B * rows * 8)) != RII_OK || (lr = c->md.ensure((size_t) B * rows * 4)) != RII_OK ||
         (lr = c->tie.ensure((size_t) B * 4)) != RII_OK || (lr = c->anyf.ensure(16)) != RII_OK || (msc && (lr = c->seq.ensure(msc)) != RII_OK))) {}
    if (lr == RII_OK && k_local > 0 && k_local < rows &&
        ((lr = c->tmp_i.ensure((size_t) B * k_local * 8)) != RII_OK || (lr = c->tmp_d.ensure((size_t) B * k_local * 4)) != RII_OK)) {}
    int64_t *rec_i = reinterpret_cast<int64_t *>(c->rec.as<unsigned char>() + kRecHeader);
    float *rec_d = reinterpret_cast<float *>(c->rec.as<unsigned char>() + kRecHeader + (size_t) B * rows * 8);
    if (const int rb = begin_on(e, st)) { c->broken = true; return rb; }      // (see rii_query_ivf_dbsharded_dev)
    int r = RII_OK;
    do {
        if (lr == RII_OK) {
            if (k_local == rows) {                 // the engine writes its LOCAL ids and distances straight into the record
                lr = query_linear_dev(e, d_queries, B, rows, S_global ? d_tids_local : nullptr, S_global ? S_local : 0, rec_i, rec_d, st);
            } else {                               // fewer local codes / targets than rows: padding rows (key 2^62, distance +inf)
                if (launch_fill_pad(rec_i, rec_d, B * rows, st) != hipSuccess) lr = set_err(RII_ERR_HIP, "fill failed");
                if (lr == RII_OK && k_local > 0) {
                    lr = query_linear_dev(e, d_queries, B, k_local, S_global ? d_tids_local : nullptr, S_global ? S_local : 0, c->tmp_i.as<int64_t>(),
                                          c->tmp_d.as<float>(), st);
                    if (lr == RII_OK && launch_copy_cols(c->tmp_i.as<int64_t>(), c->tmp_d.as<float>(), B, k_local, rows, k_local, rec_i, rec_d, st) != hipSuccess)
                        lr = set_err(RII_ERR_HIP, "copy failed");
                }
            }
        }
        const std::string local_msg = g_err;
        if (comm_set_header(c, id_offset, lr != RII_OK ? 1 : 0, st) != RII_OK) { c->broken = true; r = RII_ERR_HIP; break; }
        if ((r = comm_gather(c, c->rec.p, c->gathered.p, stride, st)) != RII_OK) break;
        if (lr != RII_OK) { g_err = local_msg; r = lr; break; }         // the peers learn it from this rank's header
        if (d_out_overflow && hipMemsetAsync(d_out_overflow, 0, (size_t) B * sizeof(int32_t), st) != hipSuccess) { r = set_err(RII_ERR_HIP, "memset failed"); break; }
        if (topk == 1) {
            if (d_out_tie && hipMemsetAsync(d_out_tie, 0, (size_t) B * sizeof(int32_t), st) != hipSuccess) { r = set_err(RII_ERR_HIP, "memset failed"); break; }
            if (launch_merge_top1(c->gathered.p, G, B, nullptr, d_out_ids, d_out_dists, st, kRecHeader) != hipSuccess) r = set_err(RII_ERR_HIP, "merge failed");
            break;
        }
        int32_t *d_tie = d_out_tie ? d_out_tie : c->tie.as<int32_t>();
        if (hipMemsetAsync(c->anyf.p, 0, 4, st) != hipSuccess) { r = set_err(RII_ERR_HIP, "memset failed"); break; }
        if (launch_merge_topk(c->gathered.p, G, B, rows, rows, 0, c->mi.as<int64_t>(), c->md.as<float>(), nullptr, st, nullptr, rows,
                              d_tie, c->anyf.as<int32_t>(), kRecHeader, c->seq.p) != hipSuccess ||
            launch_copy_cols(c->mi.as<int64_t>(), c->md.as<float>(), B, rows, topk, topk, d_out_ids, d_out_dists, st) != hipSuccess) { r = set_err(RII_ERR_HIP, "merge failed"); break; }
        int32_t h_any = 0;                     // the batch's one host read
        if (hipMemcpyAsync(&h_any, c->anyf.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { r = set_err(RII_ERR_HIP, "copy failed"); break; }
        if (h_any & 2) { r = set_err(RII_ERR_STATE, "a peer rank failed this sharded call (its rows are poisoned: ids -2, distances NaN)"); break; }
        if (!h_any) break;
        // ---- exact ties across the shards: replay (identical decisions on every rank: the flags come from identical merges) ----
        if (e->QT == 0 ? !linear_tie_chunked_topk_ok(topk) : !linear_tie_chunked_supported(e->M, e->Ks, topk)) {
            // a heap deeper than the replay kernels walk (topk > 1024): the flagged queries keep the (distance, id) order among exactly
            // tied distances and say so, like a candidate list above tie_cap (d_out_overflow; the shape is the same on every rank)
            if (d_out_overflow && hipMemcpyAsync(d_out_overflow, d_tie, (size_t) B * sizeof(int32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
                r = set_err(RII_ERR_HIP, "copy failed");
            break;
        }
        std::vector<int32_t> h_tie((size_t) B), h_sel;
        if (hipMemcpy(h_tie.data(), d_tie, (size_t) B * 4, hipMemcpyDeviceToHost) != hipSuccess) { r = set_err(RII_ERR_HIP, "copy failed"); break; }
        for (int64_t b = 0; b < B; ++b) if (h_tie[(size_t) b]) h_sel.push_back((int32_t) b);
        const int nf = (int) h_sel.size();
        const int cap = tie_cap > 0 ? tie_cap : 12288;
        if ((int64_t) G * cap >= ((int64_t) 1 << 32)) { r = set_err(RII_ERR_INVALID, "tie_cap too large"); break; }      // (same arguments on every rank)
        const int D = e->M * e->Ds;
        const size_t rec2 = linear_tie_record_bytes(nf, cap);
        int l2 = RII_OK;                       // rank-local again, agreed on before the second all-gather
        if ((l2 = c->fsel.ensure((size_t) nf * 4)) != RII_OK || (l2 = c->qf.ensure((size_t) nf * D * 4)) != RII_OK || (l2 = c->bound.ensure((size_t) nf * 4)) != RII_OK ||
            (l2 = c->rec2.ensure(rec2)) != RII_OK || (l2 = c->gg.ensure(rec2 * (size_t) G)) != RII_OK || (l2 = c->r_i.ensure((size_t) nf * topk * 8)) != RII_OK ||
            (l2 = c->r_d.ensure((size_t) nf * topk * 4)) != RII_OK) {}
        unsigned char *r2p = c->rec2.as<unsigned char>();
        const size_t cnt_bytes = ((size_t) nf * 4 + 7) / 8 * 8;
        if (l2 == RII_OK && (hipMemcpyAsync(c->fsel.p, h_sel.data(), (size_t) nf * 4, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
            l2 = set_err(RII_ERR_HIP, "copy failed");
        if (l2 == RII_OK && launch_tie_prepare(c->gathered.p, c->rank, B, rows, topk, c->fsel.as<int32_t>(), nf, d_queries, D, c->qf.as<float>(), c->bound.as<float>(), st,
                                               kRecHeader) != hipSuccess) l2 = set_err(RII_ERR_HIP, "launch failed");
        if (l2 == RII_OK && hipMemsetAsync(r2p, 0, rec2, st) != hipSuccess) l2 = set_err(RII_ERR_HIP, "memset failed");
        // a rank whose share of the target ids is EMPTY contributes nothing (S = 0 would mean "no target set" to the engine)
        if (l2 == RII_OK && !(S_global != 0 && S_local == 0))
            l2 = linear_tie_emit_locked(e, c->qf.as<float>(), nf, topk, S_global ? d_tids_local : nullptr, S_global ? S_local : 0, c->bound.as<float>(), id_offset, cap,
                                        reinterpret_cast<int64_t *>(r2p + cnt_bytes), reinterpret_cast<float *>(r2p + cnt_bytes + (size_t) nf * cap * 8),
                                        reinterpret_cast<int32_t *>(r2p), st);
        if ((r = comm_agree(c, l2, st)) != RII_OK) break;
        if ((r = comm_gather(c, r2p, c->gg.p, rec2, st)) != RII_OK) break;
        if (launch_linear_shard_replay(c->gg.p, G, nf, cap, topk, c->r_i.as<int64_t>(), c->r_d.as<float>(), st) != hipSuccess ||
            launch_tie_scatter(c->gg.p, G, nf, cap, topk, c->fsel.as<int32_t>(), c->r_i.as<int64_t>(), c->r_d.as<float>(), d_out_ids, d_out_dists, d_out_overflow, st) != hipSuccess)
            r = set_err(RII_ERR_HIP, "replay launch failed");
    } while (0);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

// Database sharding, inverted index: the protocol of ivfshard.hip driven from C (round 4; rii_amd/dist.py used to).  Coarse centres
// replicated (rii_set_coarse_centers), posting lists over this rank's codes.  (1) the per-rank list lengths after the batch's target-id
// filter are all-gathered (nlist int32 per rank); (2) every rank replays the reference's global walk on them and scores the candidates
// it owns (k + 1 rows per query); (3) ONE all-gather of (position, global id, distance) rows, merged under (distance, position);
// (4) queries whose k + 1 best distances tie exactly (d_out_tie) are redone with ALL candidates gathered (rows = L) and
// std::partial_sort replayed on the rebuilt sequence.  top-1 is asynchronous; top-k reads one word per batch on the host.
//
// Round 5: any L and any topk <= L.  The reference's own billion-scale run uses L = N / nlist = sqrt(N) ~ 31.6 k
// (examples/benchmark/run_sift1b.py:105-106): the shard kernel selects its k + 1 rows through an LDS buffer with a running bound
// (ivf_shard_any_kernel), the replay of (4) rebuilds sequences of any length in global scratch (shard_replay_any_kernel), in groups of
// flagged queries that keep the gathered rows under kShardGatherBudget.  k + 1 rows beyond what a launch can select (> 6144 above L = 8192):
// the collect-all route -- every rank sends EVERY candidate it owns (rows = L, by position), in groups of queries, and
// std::partial_sort replayed on the rebuilt sequence IS the reference's answer (no merge, no flags).
// Failures: see rii_query_linear_dbsharded_dev.
constexpr size_t kShardGatherBudget = (size_t) 512 << 20;

RII_API int rii_query_ivf_dbsharded_dev(rii_engine *e, rii_comm *c, int64_t id_offset, int64_t N_global, const float *d_queries, int64_t B,
                                        int topk, const int64_t *d_tids_local, int64_t S_local, int64_t S_global, int64_t L,
                                        int64_t *d_out_ids, float *d_out_dists, int64_t *d_out_counts, int32_t *d_out_tie, void *stream)
{
    if (!e || !c || B < 0 || S_local < 0 || S_global < 0 || (S_local > 0 && !d_tids_local) || id_offset < 0 || N_global < 1 ||
        (B > 0 && (!d_queries || !d_out_ids || !d_out_dists || !d_out_counts)) || (S_global == 0 && S_local != 0))
        return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> gc(c->mu);
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(comm_usable(c));
    const int G = c->G;
    const int64_t nlist = nlist_of(e);
    // (the shape -- M, Ks, nlist, the centres -- is the same on every rank by contract, and so is everything derived from it)
    int64_t w = 0;
    RII_TRY(ivf_shard_check(e, B, topk, S_global, L, N_global, L, &w));                    // (rows = L is always served)
    const int64_t k1 = (int64_t) topk + 1;
    const bool collect_all = k1 > (int64_t) ivf_shard_max_select_rows(e->M, e->Ks, (int) nlist, L, w) || (int64_t) G * k1 >= ((int64_t) 1 << 31);
    if (B == 0) return RII_OK;
    hipStream_t st = stream ? (hipStream_t) stream : e->stream;
    const int D = e->M * e->Ds;
    const size_t lens_bytes = (size_t) nlist * sizeof(int32_t);
    if (c->lens.ensure((size_t) (G + 1) * lens_bytes + 64) != RII_OK) {
        c->broken = true;
        return set_err(RII_ERR_HIP, "no memory for the exchange records (the communicator is unusable: the peers wait in their all-gather)");
    }
    int32_t *glen = c->lens.as<int32_t>(), *mylen = glen + (size_t) G * nlist;
    // groups of queries whose gathered every-candidate rows (G x L x 20 bytes each) stay under the budget: the exact-tie replay, and
    // the whole batch on the collect-all route
    const int64_t per_q_all = (int64_t) G * L * 20;
    const int64_t group = std::max<int64_t>(1, std::min<int64_t>(B, (int64_t) kShardGatherBudget / std::max<int64_t>(per_q_all, 1)));
    // (a rank that cannot even order its stream leaves the call before its all-gathers -- the communicator is marked
    //  unusable, as for a failed collective: include/rii_amd.h)
    if (const int rb = begin_on(e, st)) { c->broken = true; return rb; }
    int r = RII_OK;
    do {
        int lr = RII_OK;                       // rank-local outcome; the collectives below are issued regardless
        // (1) list lengths of every rank (a rank that failed sends zeros: an empty shard, a consistent walk on every peer)
        // (round 6: gathered straight from the engine's own array -- a device-to-device copy into the send slot was a launch per batch)
        const int32_t *lens_src = nullptr;
        lr = ivf_list_lengths_locked(e, d_tids_local, S_local, S_global, nullptr, st, &lens_src);
        if (lr != RII_OK) {
            if (hipMemsetAsync(mylen, 0, lens_bytes, st) != hipSuccess) { c->broken = true; r = RII_ERR_HIP; break; }
            lens_src = mylen;
        }
        std::string local_msg = g_err;
        if ((r = comm_gather(c, lens_src, glen, lens_bytes, st)) != RII_OK) break;
        if (collect_all) {
            // ---- every candidate of every query, group by group; the replay is the answer ----
            const size_t nr = (size_t) group * (size_t) L, rec2 = merge_record_bytes(group, (int) L, 1);
            if (lr == RII_OK &&
                ((lr = c->tmp_i.ensure(nr * 8)) != RII_OK || (lr = c->tmp_d.ensure(nr * 4)) != RII_OK || (lr = c->qf.ensure(nr * 4 + (size_t) group * 4)) != RII_OK ||
                 (lr = c->rec2.ensure(rec2)) != RII_OK || (lr = c->gg.ensure(rec2 * (size_t) G)) != RII_OK ||
                 (lr = c->seq.ensure(std::max<size_t>(shard_replay_scratch(group, (int) L), 16))) != RII_OK)) {}
            if (lr != RII_OK) local_msg = g_err;
            g_err = local_msg;
            if ((r = comm_agree(c, lr, st)) != RII_OK) break;
            if (d_out_tie && hipMemsetAsync(d_out_tie, 0, (size_t) B * sizeof(int32_t), st) != hipSuccess) { r = set_err(RII_ERR_HIP, "memset failed"); break; }
            for (int64_t b0 = 0; b0 < B && r == RII_OK; b0 += group) {
                const int64_t nb = std::min<int64_t>(group, B - b0);
                const size_t n = (size_t) nb * (size_t) L, recb = merge_record_bytes(nb, (int) L, 1);
                int32_t *pos = c->qf.as<int32_t>(), *nloc = pos + n;
                int l2 = ivf_shard_locked(e, d_queries + b0 * D, nb, topk, d_tids_local, S_local, S_global, L, w, glen, G, c->rank, (int) L, c->tmp_i.as<int64_t>(),
                                          c->tmp_d.as<float>(), pos, nloc, d_out_counts + b0, st);
                if (l2 == RII_OK && launch_ivf_pack(c->tmp_i.as<int64_t>(), pos, c->tmp_d.as<float>(), (int64_t) n, id_offset, c->rec2.p, st) != hipSuccess)
                    l2 = set_err(RII_ERR_HIP, "pack failed");
                if ((r = comm_agree(c, l2, st)) != RII_OK) break;
                if ((r = comm_gather(c, c->rec2.p, c->gg.p, recb, st)) != RII_OK) break;
                if (launch_shard_replay(c->gg.p, G, nb, (int) L, topk, d_out_ids + b0 * topk, d_out_dists + b0 * topk, c->seq.p, st) != hipSuccess)
                    r = set_err(RII_ERR_HIP, "replay launch failed");
            }
            break;
        }
        // (2) this rank's k + 1 best candidates per query
        const size_t n1 = (size_t) B * (size_t) k1;
        const size_t rec_bytes = merge_record_bytes(B, (int) k1, 1), stride = rec_bytes + kRecHeader;
        if (c->rec.ensure(stride) != RII_OK || c->gathered.ensure(stride * (size_t) G) != RII_OK) {
            c->broken = true;
            r = set_err(RII_ERR_HIP, "no memory for the exchange records (the communicator is unusable: the peers wait in their all-gather)");
            break;
        }
        const size_t msc = merge_topk_scratch(G, B, (int) k1);
        if (lr == RII_OK &&
            ((lr = c->tmp_i.ensure(n1 * 8)) != RII_OK || (lr = c->tmp_d.ensure(n1 * 4)) != RII_OK || (lr = c->qf.ensure(n1 * 4 + (size_t) B * 4)) != RII_OK ||
             (lr = c->bound.ensure((size_t) B * 8)) != RII_OK || (lr = c->mi.ensure(n1 * 8)) != RII_OK || (lr = c->md.ensure(n1 * 4)) != RII_OK ||
             (lr = c->r_i.ensure(n1 * 8)) != RII_OK || (lr = c->tie.ensure((size_t) B * 4)) != RII_OK || (lr = c->anyf.ensure(16)) != RII_OK ||
             (msc && (lr = c->seq.ensure(msc)) != RII_OK))) {}
        int32_t *pos = c->qf.as<int32_t>(), *nloc = pos + n1;
        int64_t *cnt = c->bound.as<int64_t>();
        // (round 6: the shard kernel writes the exchange record itself -- one launch less per batch than a pack kernel behind it)
        if (lr == RII_OK)
            lr = ivf_shard_locked(e, d_queries, B, topk, d_tids_local, S_local, S_global, L, w, glen, G, c->rank, (int) k1, c->tmp_i.as<int64_t>(),
                                  c->tmp_d.as<float>(), pos, nloc, cnt, st, c->rec.as<unsigned char>() + kRecHeader, id_offset, c->anyf.as<int32_t>());
        // (3) one all-gather + merge under (distance, position), the global ids as payload
        int32_t *d_tie = d_out_tie ? d_out_tie : c->tie.as<int32_t>();
        if (lr != RII_OK) local_msg = g_err;
        if (comm_set_header(c, 0, lr != RII_OK ? 1 : 0, st) != RII_OK) { c->broken = true; r = RII_ERR_HIP; break; }
        if ((r = comm_gather(c, c->rec.p, c->gathered.p, stride, st)) != RII_OK) break;
        if (lr != RII_OK) { g_err = local_msg; r = lr; break; }          // the peers learn it from this rank's header
        // (c->anyf's two words were cleared by block 0 of the shard kernel: round 6, a memset launch less per batch)
        // (keys = positions -> r_i, payload = ids -> mi; the merge's own OR of the flags goes to the second word: the finishing kernel
        //  recomputes it over the queries that were found)
        if (topk == 1) {                      // round 6: merge + finish of a top-1 batch in one thread-per-query launch
            if (launch_ivf_merge_top1(c->gathered.p, G, B, kRecHeader, cnt, d_out_ids, d_out_dists, d_out_counts, d_tie, c->anyf.as<int32_t>(), st) != hipSuccess)
                r = set_err(RII_ERR_HIP, "merge failed");
            break;
        }
        if (launch_merge_topk(c->gathered.p, G, B, (int) k1, (int) k1, 1, c->r_i.as<int64_t>(), c->md.as<float>(), c->mi.as<int64_t>(), st, nullptr, (int) k1, d_tie,
                              c->anyf.as<int32_t>() + 1, kRecHeader, c->seq.p) != hipSuccess ||
            launch_ivf_finish(c->mi.as<int64_t>(), c->md.as<float>(), cnt, B, (int) k1, topk, d_out_ids, d_out_dists, d_out_counts, d_tie, c->anyf.as<int32_t>(), st) != hipSuccess) {
            r = set_err(RII_ERR_HIP, "merge failed");
            break;
        }
        if (topk == 1) break;
        int32_t h_any[2] = {0, 0};             // the batch's one host read
        if (hipMemcpyAsync(h_any, c->anyf.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { r = set_err(RII_ERR_HIP, "copy failed"); break; }
        if (h_any[1] & 2) { r = set_err(RII_ERR_STATE, "a peer rank failed this sharded call (its rows are poisoned: ids -2, distances NaN)"); break; }
        if (!h_any[0]) break;
        // (4) exact ties among the k + 1 best: every rank sends ALL the candidates it owns for those queries, the sequence is rebuilt by
        //     position and std::partial_sort (src/rii.h:312-313) replayed on it -- identically on every rank, group by group
        std::vector<int32_t> h_tie((size_t) B), h_sel;
        if (hipMemcpy(h_tie.data(), d_tie, (size_t) B * 4, hipMemcpyDeviceToHost) != hipSuccess) { r = set_err(RII_ERR_HIP, "copy failed"); break; }
        for (int64_t b = 0; b < B; ++b) if (h_tie[(size_t) b]) h_sel.push_back((int32_t) b);
        const int nf = (int) h_sel.size(), rows = (int) L;
        const int fgroup = (int) std::min<int64_t>(nf, group);
        const size_t nr = (size_t) fgroup * rows, rec2 = merge_record_bytes(fgroup, rows, 1);
        int l2 = RII_OK;
        if ((l2 = c->fsel.ensure((size_t) nf * 4)) != RII_OK || (l2 = c->rec2.ensure(rec2 + (size_t) fgroup * D * 4 + 64)) != RII_OK || (l2 = c->gg.ensure(rec2 * (size_t) G)) != RII_OK ||
            (l2 = c->tmp_i.ensure(nr * 8)) != RII_OK || (l2 = c->tmp_d.ensure(nr * 4)) != RII_OK || (l2 = c->qf.ensure(nr * 4 + (size_t) fgroup * 4)) != RII_OK ||
            (l2 = c->bound.ensure((size_t) fgroup * 8)) != RII_OK || (l2 = c->r_i.ensure((size_t) fgroup * topk * 8)) != RII_OK || (l2 = c->r_d.ensure((size_t) fgroup * topk * 4)) != RII_OK ||
            (l2 = c->seq.ensure(std::max<size_t>(shard_replay_scratch(fgroup, rows), 16))) != RII_OK) {}
        if (l2 == RII_OK && (hipMemcpyAsync(c->fsel.p, h_sel.data(), (size_t) nf * 4, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess))
            l2 = set_err(RII_ERR_HIP, "copy failed");
        if ((r = comm_agree(c, l2, st)) != RII_OK) break;
        for (int f0 = 0; f0 < nf && r == RII_OK; f0 += fgroup) {
            const int nfc = std::min(fgroup, nf - f0);
            const size_t nrc = (size_t) nfc * rows, recc = merge_record_bytes(nfc, rows, 1);
            float *qsel = reinterpret_cast<float *>(c->rec2.as<unsigned char>() + ((rec2 + 63) & ~(size_t) 63));
            const int32_t *fsel = c->fsel.as<int32_t>() + f0;
            int32_t *fpos = c->qf.as<int32_t>(), *fnloc = fpos + nrc;
            int l3 = RII_OK;
            if (launch_gather_rows(d_queries, fsel, nfc, D, qsel, st) != hipSuccess) l3 = set_err(RII_ERR_HIP, "launch failed");
            if (l3 == RII_OK)
                l3 = ivf_shard_locked(e, qsel, nfc, topk, d_tids_local, S_local, S_global, L, w, glen, G, c->rank, rows, c->tmp_i.as<int64_t>(), c->tmp_d.as<float>(),
                                      fpos, fnloc, c->bound.as<int64_t>(), st);
            if (l3 == RII_OK && launch_ivf_pack(c->tmp_i.as<int64_t>(), fpos, c->tmp_d.as<float>(), (int64_t) nrc, id_offset, c->rec2.p, st) != hipSuccess)
                l3 = set_err(RII_ERR_HIP, "pack failed");
            if ((r = comm_agree(c, l3, st)) != RII_OK) break;
            if ((r = comm_gather(c, c->rec2.p, c->gg.p, recc, st)) != RII_OK) break;
            if (launch_shard_replay(c->gg.p, G, nfc, rows, topk, c->r_i.as<int64_t>(), c->r_d.as<float>(), c->seq.p, st) != hipSuccess ||
                launch_scatter_rows(fsel, nfc, topk, c->r_i.as<int64_t>(), c->r_d.as<float>(), d_out_ids, d_out_dists, st) != hipSuccess)
                r = set_err(RII_ERR_HIP, "replay launch failed");
        }
    } while (0);
    const std::string msg = g_err;
    const int r2 = end_on(e, st);
    if (r != RII_OK) { g_err = msg; return r; }
    return r2;
}

RII_API int rii_dtable(rii_engine *e, const float *queries, int64_t B, float *out)
{
    if (!e || !queries || !out || B < 0) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_on(e, e->stream));
    if (B == 0) return RII_OK;
    RII_TRY(stage_inputs(e, queries, B, nullptr, 0, 1));
    RII_TRY(build_lut(e, e->s_queries.as<float>(), B, e->stream));          // tile-interleaved (QT) layout
    const size_t bytes = (size_t) B * e->M * e->Ks * sizeof(float);
    RII_TRY(e->s_keys_a.ensure(bytes));
    HIP_TRY(launch_lut_untile(e->s_lut.as<float>(), B, e->M, e->Ks, e->lut_qt, e->s_keys_a.as<float>(), e->stream));
    HIP_TRY(hipMemcpyAsync(out, e->s_keys_a.p, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RII_OK;
}

RII_API int rii_assign(rii_engine *e, const uint8_t *codes, int64_t n, int32_t *assign)
{
    if (!e || n < 0 || (n > 0 && (!codes || !assign))) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_on(e, e->stream));
    if (e->centers.empty()) return set_err(RII_ERR_STATE, "no coarse centres");
    if (n == 0) return RII_OK;
    RII_TRY(e->s_sub_codes.ensure((size_t) n * e->M));
    HIP_TRY(hipMemcpyAsync(e->s_sub_codes.p, codes, (size_t) n * e->M, hipMemcpyHostToDevice, e->stream));
    std::vector<int32_t> a;
    RII_TRY(assign_device_codes(e, e->s_sub_codes.as<uint8_t>(), n, a));
    memcpy(assign, a.data(), (size_t) n * sizeof(int32_t));
    return RII_OK;
}

RII_API int rii_fscan_lane_subspace(int M, int lane, int t)
{
    if ((M != 16 && M != 32 && M != 64) || lane < 0 || lane > 63 || t < 0 || t >= M / 4) return -1;
    return fscan_mx_subspace(M, lane, t);
}

RII_API int rii_set_option(rii_engine *e, const char *key, int64_t value)
{
    if (!e || !key) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    const std::string k(key);
    if (k == "lut_mode") {
        if (value != RII_LUT_EXACT && value != RII_LUT_MFMA) return set_err(RII_ERR_INVALID, "bad lut_mode");
        e->lut_mode = (int) value;
    } else if (k == "scan_chunks") {
        e->scan_chunks = (int) value;
    } else if (k == "timing") {
        if (value < 0 || value > 2) return set_err(RII_ERR_INVALID, "timing must be 0, 1 (every kernel) or 2 (dominant kernels only)");
        e->timing = (int) value;
    } else if (k == "scan_mode") {
        if (value != 0 && value != 1) return set_err(RII_ERR_INVALID, "scan_mode must be 0 (exact) or 1 (filter + re-rank)");
        e->scan_mode = (int) value;
    } else if (k == "ivf_fused") {
        e->ivf_fused = value ? 1 : 0;
    } else if (k == "ivf_dbg_stop") {
        e->ivf_dbg_stop = (int) value;
    } else if (k == "shard_dbg_stop") {
        e->shard_dbg_stop = (int) (value & 0xff);
    } else if (k == "shard_force_replay") {
        e->shard_force_replay = value ? 1 : 0;
    } else if (k == "shard_pre") {
        if (value < 0 || value > 2) return set_err(RII_ERR_INVALID, "shard_pre must be 0, 1 or 2");
        e->shard_pre = (int) value;
    } else if (k == "ivf_quad") {
        e->ivf_quad = value < 0 ? 0 : (int) std::min<int64_t>(value, 2);        // 2: at every batch size (tests)
    } else if (k == "ivf_rot") {
        e->ivf_rot = value < 0 ? 0 : (int) std::min<int64_t>(value, 2);
        e->rot_failed = false;
        if (!value) { e->d_rcent.release(); e->d_rlcodes.release(); e->d_rl_toff.release(); e->rot_valid = false; }
    } else if (k == "ivf_inline_exact") {
        e->ivf_inline_exact = value ? 1 : 0;
    } else if (k == "ivf_list_codes") {
        e->ivf_list_codes = value ? 1 : 0;
        e->lcodes_failed = false; e->rot_failed = false;
        if (!value) { e->d_lcodes.release(); e->lcodes_valid = false; e->d_rlcodes.release(); e->rot_valid = false; }
    } else if (k == "ivf_force_exact") {
        e->ivf_force_exact = value ? 1 : 0;
    } else if (k == "fused_tables") {
        e->fused_tables = value ? 1 : 0;
    } else if (k == "generic_table_levels") {
        if (value != 0 && value != 63 && value != 127 && value != 255) return set_err(RII_ERR_INVALID, "generic_table_levels must be 0, 63, 127 or 255");
        e->generic_levels = (int) value;
    } else if (k == "table_levels") {
        if (value != 63 && value != 127 && value != 255) return set_err(RII_ERR_INVALID, "table_levels must be 63, 127 or 255");
        e->table_levels = (int) value;
    } else if (k == "scan_order") {
        e->scan_order = value ? 1 : 0;
    } else if (k == "scan_mx") {
        HIP_TRY(hipSetDevice(e->device));
        RII_TRY(begin_exclusive(e));
        e->scan_mx = value ? 1 : 0;
        e->fc_cov = 0;                  // the formatted lookups are laid out per kernel
    } else if (k == "scan_pipe") {
        if (value < 0 || value > 2) return set_err(RII_ERR_INVALID, "scan_pipe must be 0 (never), 1 (batches <= 128) or 2 (always)");
        e->scan_pipe = (int) value;
    } else if (k == "scan_dual") {
        e->scan_dual = value ? 1 : 0;
    } else if (k == "slice_topk") {
        e->slice_topk = value ? 1 : 0;
    } else if (k == "host_spin") {
        e->host_spin = value ? 1 : 0;
    } else if (k == "small_topk") {
        e->small_topk = value ? 1 : 0;
    } else if (k == "host_zero_copy") {
        if (value < 0 || value > 2) return set_err(RII_ERR_INVALID, "host_zero_copy must be 0 (never), 1 (auto) or 2 (always)");
        e->host_zero_copy = (int) value;
    } else if (k == "fused_rerank") {
        e->fused_rerank = value ? 1 : 0;
    } else if (k == "lanes") {
        if (value != 1 && value != 2) return set_err(RII_ERR_INVALID, "lanes must be 1 or 2");
        HIP_TRY(hipSetDevice(e->device));
        RII_TRY(begin_exclusive(e));
        e->lanes = (int) value;
    } else if (k == "fast_min_batch") {
        e->fast_min_batch = (int) std::max<int64_t>(0, value);
    } else if (k == "cand_cap") {
        if (value < 1 || value > (1 << 20)) return set_err(RII_ERR_INVALID, "bad cand_cap");
        e->cand_cap = (int) value;
        e->cand_cap_forced = true;
    } else {
        return set_err(RII_ERR_INVALID, "unknown option '%s'", key);
    }
    return RII_OK;
}
RII_API int64_t rii_get_option(const rii_engine *e, const char *key)
{
    if (!e || !key) return -1;
    std::lock_guard<std::mutex> guard(e->mu);
    const std::string k(key);
    if (k == "lut_mode") return e->lut_mode;
    if (k == "scan_chunks") return e->scan_chunks;
    if (k == "timing") return e->timing;
    if (k == "scan_mode") return e->scan_mode;
    if (k == "cand_cap") return e->cand_cap;
    if (k == "ivf_fused") return e->ivf_fused;
    if (k == "ivf_force_exact") return e->ivf_force_exact;
    if (k == "ivf_inline_exact") return e->ivf_inline_exact;
    if (k == "ivf_quad") return e->ivf_quad;
    if (k == "ivf_rot") return e->ivf_rot;
    if (k == "ivf_rot_launches") return e->rot_launches;
    if (k == "ivf_quad_launches") return e->quad_launches;
    if (k == "shard_pre_launches") return e->shard_pre_launches;
    if (k == "shard_pre") return e->shard_pre;
    if (k == "ivf_list_codes") return e->ivf_list_codes;
    if (k == "fused_tables") return e->fused_tables;
    if (k == "table_levels") return e->table_levels;
    if (k == "generic_table_levels") return e->generic_levels;
    if (k == "scan_order") return e->scan_order;
    if (k == "lanes") return e->lanes;
    if (k == "scan_mx") return e->scan_mx;
    if (k == "scan_dual") return e->scan_dual;
    if (k == "scan_pipe") return e->scan_pipe;
    if (k == "small_topk") return e->small_topk;
    if (k == "fused_rerank") return e->fused_rerank;
    if (k == "host_zero_copy") return e->host_zero_copy;
    if (k == "host_spin") return e->host_spin;
    if (k == "slice_topk") return e->slice_topk;
    if (k == "fast_min_batch") return e->fast_min_batch;
    if (k == "cand_total" || k == "cand_max") {       // debug: candidates emitted by the last filter pass (synchronises)
        if (e->last_fs_B == 0 || !e->s_cand_cnt.p) return 0;
        std::vector<unsigned int> h((size_t) e->last_fs_B);
        if (hipDeviceSynchronize() != hipSuccess) return -1;
        if (hipMemcpy(h.data(), e->s_cand_cnt.p, h.size() * sizeof(unsigned int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        int64_t tot = 0, mx = 0;
        for (unsigned int v : h) { tot += v; mx = std::max<int64_t>(mx, v); }
        return k == "cand_total" ? tot : mx;
    }
    if (k == "lut_tile") return e->QT;
    if (k == "n_cu") return e->n_cu;
    return -1;
}

RII_API int rii_timing_read(rii_engine *e, const char *kernel, double *total_ms, int64_t *launches)
{
    if (!e || !kernel) return set_err(RII_ERR_INVALID, "bad arguments");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_exclusive(e));
    KernelTimer &t = e->timers[kernel];
    for (auto &pr : t.pending) {
        HIP_TRY(hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        t.total_ms += ms;
        t.launches += 1;
        e->event_pool.push_back(pr.first);
        e->event_pool.push_back(pr.second);
    }
    t.pending.clear();
    if (total_ms) *total_ms = t.total_ms;
    if (launches) *launches = t.launches;
    return RII_OK;
}
RII_API int rii_timing_reset(rii_engine *e)
{
    if (!e) return set_err(RII_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> guard(e->mu);
    for (auto &kv : e->timers) {
        for (auto &pr : kv.second.pending) { (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second); }
        kv.second.pending.clear();
        kv.second.total_ms = 0.0;
        kv.second.launches = 0;
    }
    return RII_OK;
}
RII_API int rii_synchronize(rii_engine *e)
{
    if (!e) return set_err(RII_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> guard(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    RII_TRY(begin_exclusive(e));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RII_OK;
}
