#!/usr/bin/env python3
"""bench.py -- queries/sec + recall@1 of the batched IVFPQ query path (BASELINE.json configs[1]: SIFT1M-shaped,
D=128, M=32, Ks=256, batch=1024, top-1 linear ADC scan) on N MI355X GPUs of one node.

A step = one pass of the hot path (distance-table build + scan + top-k) over one batch of 1024 queries per GPU, inputs
and outputs resident in HBM.  ONE JSON line from rank 0:

  value / ms_per_step   the named workload, K timed steps (weak scaling: every rank owns a different 1024-query batch over a
                        replicated index; query sharding has no exchange step, so `value` carries no collective)
  roofline              the dominant kernel timed by HIP events attached to each of its dispatches inside the timed region,
                        against the resource that binds it (LDS row-gather rate for the scans, the issue/latency floor of
                        ivf_fused_kernel for the inverted index, HBM for the single-query Deep scan)
  cpu_baseline          the real reference build (oracle/_ref; else the C oracle) on this box's host cores, same index and
                        queries, with `ids_match_gpu`; rank 0, N=1 only
  others                (default invocation, N=1) EVERY other BASELINE config measured in the same process: configs[2] `ivf`,
                        configs[3] `subset` and `subset_ivf` on the same index, configs[0] `readme_n10k` (single-query latency)
                        and a configs[4]-shaped `deep_shard` (D=96, M=16, 16 M codes) -- each with ms_per_step, dominant-kernel
                        time, roofline.frac against its own binding resource, and cpu_baseline + ids_match_gpu
  strong                (N > 1) BASELINE's "batch=1024 at 1/2/4/8 GPU": the SAME global batch of 1024 split over the ranks --
                        `query_sharded` (index replicated, rows all-gathered inside the timed region) and `db_sharded` (codes
                        split N ways, every rank answers the whole batch on its shard: all-gather + device merge inside the
                        timed region, through rii_amd.dist.DbShardedIndex)
  with_gather           (N > 1, or N = 1 under torchrun) the weak loop with the RCCL all-gather of the result rows inside
  host_call / uninstrumented / fresh_queries / pipelined: the same step through the host-pointer C ABI; without any timing
                        event; over 8 distinct batches; alternating on two HIP streams.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N ranks
(`python -m torch.distributed.run --nproc-per-node N`, rendezvous on 127.0.0.1); under torchrun it is one rank.
`--workload deep`: configs[4] shape, --n-base codes PER GPU, database sharded, through DbShardedIndex (device merge, no host sync).
`--latency`: one call per step with fresh queries and a synchronisation per call (the reference's usage pattern).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LDS_PEAK_GBPS = 157286.4         # ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X guide, LDS table)
HBM_PEAK_GBPS = 8000.0
N_CU, N_SIMD, CLK_GHZ = 256, 1024, 2.4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n-base", type=int, default=1_000_000)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--workload", default="linear", choices=["linear", "ivf", "subset", "subset-ivf", "deep", "deep-ivf"],
                    help="linear/ivf/subset/subset-ivf: BASELINE configs[1..3] (SIFT1M-shaped, index replicated, queries "
                         "sharded); deep: configs[4] shape (D=96, M=16, --n-base codes PER GPU, database sharded); deep-ivf: the same "
                         "shards searched through the database-sharded inverted index (nlist = sqrt(N_global), L = N_global / nlist: "
                         "the reference's billion-scale setting, examples/benchmark/run_sift1b.py:105-106)")
    ap.add_argument("--nlist", type=int, default=0, help="inverted-index workloads: coarse lists (0 = 1024; deep-ivf: sqrt(N_global))")
    ap.add_argument("--L", type=int, default=0, help="inverted-index workloads: candidates per query (0 = L0 = round(N / nlist))")
    ap.add_argument("--topk", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-call", action="store_true")
    ap.add_argument("--no-fresh", action="store_true",
                    help="skip the extra measurement with a different query batch every step")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="skip the rocprofv3 --pmc passes of a short copy of this run (HBM traffic and LDS counters of the dominant "
                         "kernel, measured live on rank 0 at N = 1; without them the committed profiles/*.json tables are quoted)")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra two-stream measurement (reported beside `value`, never as it)")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the `others` object (the other BASELINE configs measured in the same process)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling measurements")
    ap.add_argument("--deep-shard", type=int, default=64_000_000,
                    help="codes of the `others.deep_shard` measurement (default 64 M codes = 1 GB: four times the 256 MB Infinity Cache, so "
                         "the code stream really comes from HBM; 125000000 = the per-GPU shard of Deep1B over 8 GPUs)")
    ap.add_argument("--deep-structured", type=int, default=10_000_000,
                    help="`others.deep_structured`: vectors of the structured Deep1B-shaped set (0 = skip)")
    ap.add_argument("--preheat", type=float, default=0.15,
                    help="seconds of untimed steps before the W warm-up steps (clock ramp; 0 under rocprofv3 counter passes)")
    ap.add_argument("--latency", action="store_true",
                    help="per-call latency: fresh queries every call, one synchronisation per call (use with --batch 1)")
    ap.add_argument("--lut-mode", default="exact", choices=["exact", "mfma"])
    ap.add_argument("--shard-dbg-stop", type=int, default=0,
                    help="measurement only (--workload deep-ivf): ivf_shard_any_kernel returns after this phase; the rows are wrong then")
    ap.add_argument("--scan-order", type=int, default=1, choices=[0, 1],
                    help="1: scan the LDS-friendly permutation of the codes (default), 0: id order")
    ap.add_argument("--scan-mx", type=int, default=1, choices=[0, 1],
                    help="1 (default): the filter scan of the M = 16 / 32 shapes sums its table bytes on the matrix cores "
                         "(fscan_mx_kernel); 0: on the vector ALU (fscan_kernel)")
    ap.add_argument("--full-out", default="",
                    help="where the long form of the JSON line is written (default gpurun_out/bench_full_<workload>.json); the line "
                         "printed on stdout is the compact form")
    ap.add_argument("--scan-mode", type=int, default=1, choices=[0, 1],
                    help="1: 8-bit filter + exact re-rank (default), 0: exact scan of every code; identical results")
    return ap.parse_args()


def emit(line, args):
    """Rank 0's output: the long form of the line goes to a side file (--full-out; default gpurun_out/bench_full_<workload>.json
    under the repo), the line printed LAST on stdout is its compact form (rii_amd/benchline.py: <= 12 KB -- round 5's 20.6 KB line
    came back from the driver unparsed)."""
    from rii_amd import benchline
    path = args.full_out or os.path.join(ROOT, "gpurun_out", "bench_full_%s.json" % args.workload.replace("-", "_"))
    shown = None
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(line, f)
            f.write("\n")
        shown = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
    except OSError:
        pass                                     # a read-only tree must not cost the run its line
    # RCCL prints its version banner through C stdio, which is fully buffered when stdout is a pipe: left alone it comes out at
    # process exit, AFTER the JSON line.  Flushed here, the JSON line is the last line of stdout.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                            # noqa: BLE001
        pass
    print(benchline.dumps(line, shown), flush=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` typed directly: become the launcher of N ranks (what the driver does with torchrun)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


# --------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference itself (oracle/_ref) on the host cores, same index, same queries
# --------------------------------------------------------------------------------------------------------------------
def cpu_baseline(kind_of_search, what, ref_engine_factory, queries, topk, tids, L, budget_s=6.0, thread_settings=None):
    """The reference's own path, one query per call (it has no batch entry point): QueryLinear is OpenMP-parallel over
    N (src/rii.h:213,222), QueryIvf single-threaded (rii.h:244-326).  OpenMP's default thread count (all hardware
    threads) is not the reference's best setting on a many-core host, so a few counts are timed on a bounded sample and
    the BEST is reported with `cores` = the threads it used.  Returns (object, list of id lists of the longest sample)."""
    import ctypes
    ref_e, kind, flav = ref_engine_factory()
    E = np.array([], np.int64)
    tids = E if tids is None else tids
    ivf = kind_of_search == "ivf"

    def call(q):
        return ref_e.query_ivf(q, topk, tids, L) if ivf else ref_e.query_linear(q, topk, tids)

    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    ncpu = os.cpu_count() or 1
    if thread_settings is None:
        thread_settings = [ncpu, 64, 16, 8]
    settings = [1] if ivf else (sorted({min(ncpu, t) for t in thread_settings}, reverse=True) if gomp else [ncpu])
    tried, best, ids_full = [], None, None
    for nthr in settings:
        if gomp:
            gomp.omp_set_num_threads(int(nthr))
        call(queries[0])                                              # warm-up
        t0 = time.perf_counter()
        for q in queries[:4]:
            call(q)
        per = (time.perf_counter() - t0) / 4
        n = int(min(len(queries), max(8, budget_s / max(per, 1e-6))))     # ~budget_s of CPU work per setting
        t0 = time.perf_counter()
        res = [call(q)[0] for q in queries[:n]]
        dt = time.perf_counter() - t0
        tried.append("%d threads: %.1f q/s over %d queries" % (nthr, n / dt, n))
        if ids_full is None or len(res) > len(ids_full):
            ids_full = res
        if best is None or n / dt > best[0]:
            best = (n / dt, nthr, n)
    return {"value": best[0], "unit": "queries/s", "ms_per_query": 1e3 / best[0], "cores": best[1], "kind": kind,
            "sample": "one query per call, %s, top-%d%s; thread counts tried: %s"
                      % (what, topk, (", build flavour " + flav) if flav else "", "; ".join(tried))}, ids_full


def reference_factory(eng, cw, codes, arch_hint, ivf):
    """-> callable building the CPU engine: the real reference (kind "reference") when oracle/_ref loads, else the C oracle
    ("port").  For the inverted index it receives the GPU engine's centres and posting lists through its own py::pickle
    set-state (src/main.cpp:39-52)."""
    def make():
        from oracle import oracle as O
        ref, arch, flav = O.load_reference()
        if ref is not None:
            if ivf:
                e = ref.RiiCpp.__new__(ref.RiiCpp)
                e.__setstate__(eng.__getstate__())
            else:
                e = ref.RiiCpp(cw, False)
                e.add_codes(codes, False)
            return e, "reference", flav
        e = O.OracleRii(cw, False, simd_arch=arch_hint)
        e.add_codes(codes, False)
        if ivf:
            e.centers = np.array(eng.coarse_centers, np.uint8)
            e._lists = eng.posting_lists
        return e, "port", None
    return make


def ids_match(res_ids, res_cnt, cpu_res):
    n = len(cpu_res)
    if res_cnt is not None:
        return bool(all(list(res_ids[b, :int(res_cnt[b])]) == list(cpu_res[b]) for b in range(n))), n
    return bool(all(list(res_ids[b]) == list(cpu_res[b]) for b in range(n))), n


# --------------------------------------------------------------------------------------------------------------------
# roofline bookkeeping
# --------------------------------------------------------------------------------------------------------------------
def profile_table(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def live_counters(kernel_sub, shape_args, budget_s=75.0):
    """HBM traffic and LDS counters of the dominant kernel measured in THIS run: a short copy of the same command (3 steps, no side
    measurements) under `rocprofv3 --pmc`, one pass per counter group as the MI355X guide prescribes (no trace domain next to --pmc).
    Returns {} when rocprofv3 is missing, a pass fails or the budget runs out -- the caller then quotes the committed tables."""
    import csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return {}
    groups = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"))
    out, t_end = {}, time.perf_counter() + budget_s
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)                                      # the copy is a plain one-process run, whatever launched this one
    for grp in groups:
        left = t_end - time.perf_counter()
        if left < 10.0:
            break
        d = tempfile.mkdtemp(prefix="rii_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", *grp, "--kernel-include-regex", kernel_sub, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--preheat", "0", "--no-cpu-baseline",
               "--no-host-call", "--no-others", "--no-fresh", "--no-pipelined", "--no-live-counters"] + list(shape_args)
        try:
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=left)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)             # the group this call started, nothing else
                pr.wait()
                shutil.rmtree(d, ignore_errors=True)
                break
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            per = {}
            for f in files:
                for r in csv.DictReader(open(f)):
                    if kernel_sub in r["Kernel_Name"]:
                        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
            for c, dv in per.items():
                out[c] = sum(dv.values()) / len(dv)           # per launch: mean over the dispatches, instances summed
                out["_dispatches"] = len(dv)
        except Exception:
            pass
        shutil.rmtree(d, ignore_errors=True)
    return out


def filter_kernel_name(scan_mx, M):
    if scan_mx and M == 16:
        return "fscan_mx_dual_kernel"          # two 16-query tiles per block (engine option scan_dual, default on)
    return "fscan_mx_kernel" if (scan_mx and M in (32, 64)) else "fscan_kernel"


def workload_key(workload, scan_mode, scan_mx, M, n_scanned, batch, topk):
    key = "%s/scan_mode=%d/M=%d/N=%d/B=%d" % (workload, scan_mode, M, n_scanned, batch)
    if scan_mode == 1 and not scan_mx:
        key += "/scan_mx=0"
    if topk != 1:
        key += "/topk=%d" % topk
    return key


def roofline_scan(kernel_name, B, n_codes, M, Ks, avg_s, launches, steps, byte_tables, fmt_bytes, pmc_key):
    """Linear scans: the binding resource is the LDS table gather -- every (query, code, m) lookup reads one table entry
    (1 byte from the filter's byte tables, 4 bytes from the exact fp32 tables) out of LDS; the code bytes themselves are
    shared by the whole batch through L2/LDS, so HBM sees ~N*M bytes per launch, not B*N*M."""
    lookups = B * n_codes * M                                  # M table gathers per (query, code)
    entry = 1 if byte_tables else 4
    achieved = lookups * entry / avg_s / 1e9 if avg_s > 0 else 0.0
    pmc = profile_table("pmc.json").get(pmc_key, {})
    traffic = profile_table("traffic.json").get(pmc_key, {}).get("hbm_bytes_per_launch")
    floor = n_codes * M * fmt_bytes + B * M * Ks * entry       # the code stream once + the tables staged
    hbm = {"algorithmic_bytes_per_launch": lookups, "compulsory_floor_bytes": floor, "traffic_bytes": traffic,
           "achieved": (traffic / avg_s / 1e9) if (traffic and avg_s > 0) else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    if hbm["achieved"] is not None:
        hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBPS
    return {"bound": "lds-gather", "kernel": kernel_name, "achieved": achieved, "peak": LDS_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / LDS_PEAK_GBPS, "traffic": traffic, "avg_launch_ms": avg_s * 1e3, "launches": launches,
            "launches_per_step": launches / max(steps, 1), "table_lookups_per_launch": lookups, "entry_bytes": entry,
            "hbm": hbm,
            # NOT measured in this run: the rocprofv3 --pmc passes of the same command, kept under profiles/ (a kernel change
            # moves `frac` above, measured live, but not these until the profiles are retaken)
            "counters_from_profiles": {"source": pmc.get("source"), "conflict_frac": pmc.get("lds_conflict_frac"),
                                       "lds_busy": pmc.get("lds_busy"), "valu_busy": pmc.get("valu_busy"),
                                       "mfma_busy": pmc.get("mfma_busy"), "shader_clock_ghz": pmc.get("shader_clock_ghz"),
                                       "wave_wait_frac": pmc.get("wave_wait_frac"), "wave_active_frac": pmc.get("wave_active_frac")},
            "note": "achieved = table-entry bytes gathered from LDS per second (B*N*M lookups x entry_bytes / kernel time, the "
                    "kernel time from HIP events on its dispatches in the timed region); peak = conflict-free ds_read_b128 "
                    "rate at the nominal 2.4 GHz"}


def roofline_hbm(kernel_name, alg_bytes, avg_s, launches, steps, traffic):
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": avg_s * 1e3, "launches": launches, "launches_per_step": launches / max(steps, 1)}


def roofline_ivf(B, nlist, M, Ks, Ds, w, mean_len, L, avg_s, launches, steps, traffic):
    """ivf_fused_kernel's working set (codebook 128 KiB, centres nlist*M, the visited lists' codes) lives in L2 / Infinity Cache,
    so HBM does not bind it; neither does the LDS array (profiles/r02_ivf_pmc.json: 28 % busy).  What the kernel has to issue
    per query is fixed by the algorithm: M*Ks table entries built (Ds subtract + Ds multiply + Ds-1 add each), (nlist + L) * M
    table lookups with one fp32 add each, i.e. a floor of VALU instruction issues; `achieved` = those issues per second against
    the chip's VALU issue rate (one wave64 instruction per 4 cycles per SIMD).  The HBM-form figure of SURVEY 8(d) is kept in
    `hbm_form` for continuity."""
    per_q = M * Ks * (3 * Ds - 1) + (nlist + L) * M * 2          # lane-operations: table build + (address, add) per lookup
    insts = B * per_q / 64.0                                    # wave64 instructions
    peak = N_SIMD * CLK_GHZ * 1e9 / 4.0                          # wave instructions per second, whole chip
    achieved = insts / avg_s if avg_s > 0 else 0.0
    alg = B * (nlist * M + w * mean_len * 4 + L * M)
    return {"bound": "valu-issue", "kernel": "ivf_fused_kernel", "achieved": achieved / 1e9, "peak": peak / 1e9,
            "unit": "G wave-instructions/s", "frac": achieved / peak, "traffic": traffic, "avg_launch_ms": avg_s * 1e3,
            "launches": launches, "launches_per_step": launches / max(steps, 1),
            "algorithmic_lane_ops_per_query": per_q,
            "hbm_form": {"algorithmic_bytes_per_launch": alg, "achieved": alg / avg_s / 1e9 if avg_s > 0 else 0.0,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (alg / avg_s / 1e9 / HBM_PEAK_GBPS) if avg_s > 0 else 0.0},
            "note": "one block per query: table build + coarse scoring + candidate scan are dependent phases of ~2.7 k vector "
                    "instructions per wave; the floor counted here is the arithmetic the algorithm cannot avoid (no address "
                    "arithmetic, no selection): frac = that floor over the chip's issue rate"}


def timed_loop(fn, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    return time.perf_counter() - t0


DOMINANT = ("scan", "ivf_fused", "ivf_scan", "ivf_shard", "shard_coarse")
OTHER_KERNELS = ("lut", "quant", "rerank", "kth", "tie", "select", "gather", "ivf_exact", "ivf_coarse", "ivf_plan", "ivf_scan",
                 "ivf_select")


PREHEAT_S = 0.15


def preheat(step, sync, seconds=None):
    """The MI355X ramps its shader clock over the first ~30 ms of sustained load (tools/clock_ramp.py, profiles/r03_clock_ramp.json:
    0.415 -> 0.381 -> 0.366 -> 0.359 -> 0.356 ms per step over consecutive 20-step loops from idle, and back up after 1 s of
    idling), so W = 5 warm-up steps (2 ms) followed by K = 20 timed steps (8 ms) would time the ramp, not the kernel.  Untimed steps
    for `seconds` of wall time bring the chip to its steady state first; the W warm-up steps and the K timed steps follow as the
    contract says.  Returns the number of steps issued."""
    n = 0
    seconds = PREHEAT_S if seconds is None else min(seconds, PREHEAT_S)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            step()
        sync()
        n += 8
    return n


def preheat_count(step, sync, count):
    """The same by step count: for steps that hold a collective (every rank must issue the same number of them)."""
    if PREHEAT_S <= 0:
        return 0
    for _ in range(count):
        step()
    sync()
    return count


def measure(eng, step, steps, warmup, barrier, sync, heat=True):
    """pre-heat (clock ramp), W warm-up steps, K timed steps with ONLY the dominant kernel carrying HIP events (attached to its
    dispatch: engine option `timing` = 2), then a short untimed pass with events around every launch for the other kernels' shares."""
    if heat is True:
        preheat(step, sync)
    elif heat:
        preheat_count(step, sync, int(heat))
    for _ in range(warmup):
        step()
    barrier()
    eng.set_option("timing", 2)
    eng.timing_reset()
    elapsed = timed_loop(step, steps, barrier)
    eng.set_option("timing", 0)
    dom = {kn: eng.timing_read(kn) for kn in DOMINANT}
    eng.timing_reset()
    eng.set_option("timing", 1)
    n_break = max(3, min(steps, 10))
    for _ in range(n_break):
        step()
    sync()
    eng.set_option("timing", 0)
    shares = {}
    for kn in OTHER_KERNELS:
        ms_, n_ = eng.timing_read(kn)
        if n_:
            shares[kn + "_ms_per_step"] = ms_ / n_break
    eng.timing_reset()
    return elapsed, dom, shares


# --------------------------------------------------------------------------------------------------------------------
# `others`: the other BASELINE configs, measured in the same process (rank 0, N = 1)
# --------------------------------------------------------------------------------------------------------------------
def sift_workload(name, eng, args, torch, dev, stream, q_dev, cw, codes, barrier, arch, N, M, Ks, Ds, B, topk):
    """One of configs[2] / configs[3] on the SIFT1M-shaped index the main measurement used: `ivf`, `subset`, `subset_ivf`."""
    ivf = name in ("ivf", "subset_ivf")
    S, L, d_tids, h_tids = 0, 0, 0, None
    nlist = 1024
    if ivf:
        if eng.nlist != nlist:
            eng.reconfigure(nlist, 5)
        L = int(np.round(N / nlist))
    if name in ("subset", "subset_ivf"):
        h_tids = np.sort(np.random.default_rng(7).choice(N, min(100_000, N), replace=False)).astype(np.int64)
        tids = torch.from_numpy(h_tids).to(dev)
        S, d_tids = tids.numel(), tids.data_ptr()
    out_ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
    out_d = torch.empty((B, topk), dtype=torch.float32, device=dev)
    out_cnt = torch.empty((B,), dtype=torch.int64, device=dev)

    def step():
        if ivf:
            eng.query_ivf_dev(q_dev.data_ptr(), B, topk, d_tids, S, L, out_ids.data_ptr(), out_d.data_ptr(), out_cnt.data_ptr(), stream)
        else:
            eng.query_linear_dev(q_dev.data_ptr(), B, topk, d_tids, S, out_ids.data_ptr(), out_d.data_ptr(), stream)

    nq0, nr0 = eng.get_option("ivf_quad_launches"), eng.get_option("ivf_rot_launches")
    elapsed, dom, shares = measure(eng, step, args.steps, max(args.warmup, 2), barrier, torch.cuda.synchronize)
    plain = timed_loop(step, args.steps, barrier)          # the same K steps without the two events on the dominant kernel's dispatch
    res_ids = out_ids.cpu().numpy().copy()
    res_cnt = out_cnt.cpu().numpy().copy() if ivf else None
    k_ms, k_n = dom["ivf_fused" if ivf else "scan"]
    avg_s = (k_ms / max(args.steps, 1)) * 1e-3
    n_scanned = S if name == "subset" else N
    if ivf:
        w = min(nlist, int(np.round(L * nlist / (S if S else N))) + 3)
        key = workload_key(name.replace("_", "-"), args.scan_mode, args.scan_mx, M, N, B, topk)
        roof = roofline_ivf(B, nlist, M, Ks, Ds, w, N // nlist, L, avg_s, k_n, args.steps,
                            profile_table("traffic.json").get(key, {}).get("hbm_bytes_per_launch"))
        # (the engine times its one-launch inverted-index kernels under one name: which of them served the timed steps)
        if eng.get_option("ivf_quad_launches") > nq0:
            roof["kernel"] = "ivf_quad_kernel"
        elif eng.get_option("ivf_rot_launches") > nr0:
            roof["kernel"] = "ivf_rot_kernel"
    else:
        filt = bool(args.scan_mode and (topk > 1 or B >= eng.get_option("fast_min_batch")))
        key = workload_key(name, args.scan_mode, args.scan_mx, M, n_scanned, B, topk)
        roof = roofline_scan(filter_kernel_name(args.scan_mx, M) if filt else "scan_kernel", B, n_scanned, M, Ks, avg_s, k_n,
                             args.steps, filt, 1 if args.scan_mx or not filt else 2, key)
    roof.update(shares)
    obj = {"config": "SIFT1M-shaped %s, D=128 M=%d Ks=256, N=%d, batch=%d, topk=%d%s%s"
                     % (name, M, N, B, topk, (", nlist=%d L=%d" % (nlist, L)) if L else "", (", |target_ids|=%d" % S) if S else ""),
           "value": B * args.steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / args.steps * 1e3,
           "uninstrumented_ms_per_step": plain / args.steps * 1e3, "uninstrumented_value": B * args.steps / plain,
           "kernel": roof["kernel"], "kernel_ms": avg_s * 1e3, "roofline": roof}
    if not args.no_cpu_baseline:
        what = {"ivf": "inverted index nlist=%d L=%d" % (nlist, L), "subset": "linear scan of %d target ids" % S,
                "subset_ivf": "inverted index nlist=%d L=%d over %d target ids" % (nlist, L, S)}[name]
        cb, cpu_res = cpu_baseline("ivf" if ivf else "linear", what, reference_factory(eng, cw, codes, arch, ivf),
                                   q_dev.cpu().numpy(), topk, h_tids, L, budget_s=3.0)
        cb["ids_match_gpu"], cb["queries_compared"] = ids_match(res_ids, res_cnt, cpu_res)
        obj["cpu_baseline"] = cb
    return obj


def sharded_world1_workload(eng, args, torch, dev, stream, t_q, B, topk, barrier, out_ids):
    """BASELINE's metric is quoted at 1 / 2 / 4 / 8 GPUs: what the multi-GPU step costs on top of the kernels is visible on ONE GPU too --
    the sharded C-ABI entry points (rii_query_linear_qsharded_dev / _dbsharded_dev: engine kernels -> ncclAllGather -> unpack / merge,
    all enqueued by one library call) over a ONE-rank RCCL communicator, at the global batch and at 128 queries (the per-GPU share of
    the 1024-query batch on 8 GPUs).  Same index (world size 1: the shard IS the database), same queries, rows checked."""
    from rii_amd import dist as rd
    comm = rd.get_comm()
    N = eng.N
    out = {"what": "one-rank RCCL communicator behind the C ABI; per step: ONE rii_query_linear_{q,db}sharded_dev call on the bench stream, "
                   "device-resident queries and rows; `plain` = rii_query_linear_dev on the same box, same loop",
           "rccl_ranks": comm.size}
    for b in (B, 128):
        if b > t_q.shape[0]:
            continue
        q = t_q[:b].contiguous()
        oi = torch.empty((b, topk), dtype=torch.int64, device=dev)
        od = torch.empty((b, topk), dtype=torch.float32, device=dev)
        ri = torch.empty((b, topk), dtype=torch.int64, device=dev)
        rdd = torch.empty((b, topk), dtype=torch.float32, device=dev)

        def plain():
            eng.query_linear_dev(q.data_ptr(), b, topk, 0, 0, ri.data_ptr(), rdd.data_ptr(), stream)

        def qsh():
            comm.query_linear_qsharded_dev(eng, q.data_ptr(), b, topk, 0, 0, oi.data_ptr(), od.data_ptr(), stream)

        def dbsh():
            comm.query_linear_dbsharded_dev(eng, 0, q.data_ptr(), b, topk, 0, 0, 0, oi.data_ptr(), od.data_ptr(), stream=stream)

        res = {}
        K = max(args.steps, 50)
        for name, fn in (("plain", plain), ("query_sharded", qsh), ("db_sharded", dbsh)):
            preheat(fn, torch.cuda.synchronize, 0.05)
            ms = min(timed_loop(fn, K, barrier) for _ in range(2)) / K * 1e3
            res[name] = {"ms_per_step": ms, "value": b / ms * 1e3, "unit": "queries/s"}
            if name != "plain":
                res[name]["rows_match_plain"] = bool(torch.equal(oi, ri) and torch.equal(od, rdd))
        res["query_sharded"]["added_us"] = (res["query_sharded"]["ms_per_step"] - res["plain"]["ms_per_step"]) * 1e3
        res["db_sharded"]["added_us"] = (res["db_sharded"]["ms_per_step"] - res["plain"]["ms_per_step"]) * 1e3
        # the inverted index the same three ways (round 5): plain rii_query_ivf_dev (ivf_fused_kernel), query-sharded (the same kernel +
        # all-gather + unpack) and DATABASE-sharded (rii_query_ivf_dbsharded_dev: list lengths -> all-gather -> ivf_shard_kernel ->
        # pack -> all-gather -> merge -> finish; a different kernel: the walk is global, so the fused one-query kernel does not apply)
        if eng.nlist > 0:
            nl = eng.nlist
            L = int(np.round(N / nl))
            oc = torch.empty((b,), dtype=torch.int64, device=dev)
            rc = torch.empty((b,), dtype=torch.int64, device=dev)

            def iplain():
                eng.query_ivf_dev(q.data_ptr(), b, topk, 0, 0, L, ri.data_ptr(), rdd.data_ptr(), rc.data_ptr(), stream)

            def iqsh():
                comm.query_ivf_qsharded_dev(eng, q.data_ptr(), b, topk, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), stream)

            def idbsh():
                comm.query_ivf_dbsharded_dev(eng, 0, N, q.data_ptr(), b, topk, 0, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), 0, stream)

            ires = {"nlist": nl, "L": L}
            for name, fn in (("plain", iplain), ("query_sharded", iqsh), ("db_sharded", idbsh)):
                preheat(fn, torch.cuda.synchronize, 0.05)
                ms = min(timed_loop(fn, K, barrier) for _ in range(2)) / K * 1e3
                ires[name] = {"ms_per_step": ms, "value": b / ms * 1e3, "unit": "queries/s"}
                if name != "plain":
                    ires[name]["rows_match_plain"] = bool(torch.equal(oi, ri) and torch.equal(od, rdd) and torch.equal(oc, rc))
                    ires[name]["added_us"] = (ms - ires["plain"]["ms_per_step"]) * 1e3
            res["ivf"] = ires
        out["batch_%d" % b] = res
    return out


def mfma_tables_workload(eng, args, torch, dev, stream, q_dev, B, M, Ks, Ds, barrier):
    """north_star: "MFMA used only for the batched (Q x Ks) x Ds sub-distance table build where it is a true dense contraction".  Engine
    option lut_mode = "mfma": lut_build_mfma_kernel (v_mfma_f32_16x16x4_f32: -2 q.c on the matrix cores, |q|^2 + |c|^2 added in fp32)
    writes the fp32 tables, the SAME filter scan and re-rank follow (tables quantised by lut_quantize_kernel).  Not the default: the
    expansion rounds differently from fvec_L2sqr (src/distance.h:117-252), so distances agree to 1e-4 relative instead of bit for bit.
    One row: the step, the table kernel's time and flop rate against the fp32 matrix peak, and the agreement with the exact mode."""
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev)
    od = torch.empty((B, 1), dtype=torch.float32, device=dev)

    def step():
        eng.query_linear_dev(q_dev.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), stream)

    step()
    torch.cuda.synchronize()
    ids_e, d_e = oi.cpu().numpy().copy(), od.cpu().numpy().copy()
    eng.set_option("lut_mode", "mfma")
    try:
        elapsed, dom, shares = measure(eng, step, args.steps, max(args.warmup, 2), barrier, torch.cuda.synchronize)
        ids_m, d_m = oi.cpu().numpy().copy(), od.cpu().numpy().copy()
    finally:
        eng.set_option("lut_mode", "exact")
    flops = 2.0 * B * M * Ks * Ds
    lut_ms = shares.get("lut_ms_per_step")
    peak = 157.3                                                  # TFLOP/s, fp32 matrix (MI355X guide)
    ach = (flops / (lut_ms * 1e-3) / 1e12) if lut_ms else None
    return {"config": "linear top-1, batch=%d, tables on the matrix cores (lut_mode=mfma), same filter scan + re-rank" % B,
            "ms_per_step": elapsed / args.steps * 1e3, "value": B * args.steps / elapsed, "unit": "queries/s",
            "table_kernel": "lut_build_mfma_kernel", "table_kernel_ms": lut_ms, "quantise_ms": shares.get("quant_ms_per_step"),
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": (ach / peak) if ach else None,
                         "flops_per_launch": flops, "traffic": None,
                         "note": "2 x B x M x Ks x Ds = %.0f MFLOP per batch: a launch-latency-sized problem, so the matrix pipe cannot be "
                                 "the bound (SURVEY 8d); counters: profiles/r05_mfma_*" % (flops / 1e6)},
            "ids_equal_exact_mode": float((ids_m == ids_e).mean()),
            "max_rel_distance_error_vs_exact_mode": float(np.max(np.abs(d_m - d_e) / np.maximum(np.abs(d_e), 1.0)))}


def modulo_lists(n, nlist):
    """A synthetic posting-list partition of n codes for the throughput legs that cannot afford the N x nlist x M assignment pass of
    a real build (64 M codes x 8 k lists = 8e15 table lookups): code i goes to list i mod nlist, ids ascending inside a list
    (src/rii.h:356-358).  The codes of those legs are uniform random bytes, so any partition is as good as the nearest-centre one for
    what is measured (the walk, the candidate scoring, the selection); the SAME lists go to the CPU baseline.  -> (off, ids) CSR."""
    rows = (n + nlist - 1) // nlist
    grid = np.arange(rows * nlist, dtype=np.int64).reshape(rows, nlist).T.reshape(-1)      # list-major: j, j + nlist, j + 2 nlist, ...
    ids = grid[grid < n].astype(np.int32)
    lens = np.full(nlist, n // nlist, np.int64)
    lens[:n % nlist] += 1
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return off, ids


def roofline_ivf_shard(kernel, B, nlist, M, L, avg_s, launches, steps, traffic=None, coarse_s=0.0):
    """Database-sharded inverted index.  HBM form, compulsory bytes only (the old form counted every query's pass over the
    L2-resident centres and the 4-byte ids the posting-order rows made unnecessary): of the L candidates of a query's walk the rows
    this rank owns (M bytes each, read by exactly one query: they stream from HBM at any shard that outgrows the caches) + the centres
    once.  `avg_s` = the walk kernel + (round 6) the coarse pre-pass of the same batch.  The centre scores -- B x nlist x M table
    gathers that never leave the chip -- are in `lds_form`."""
    alg = B * L * M + nlist * M
    r = roofline_hbm(kernel, alg, avg_s, launches, steps, traffic)
    gathers = B * (nlist + L) * M * 4
    r["lds_form"] = {"table_gather_bytes_per_launch": gathers, "achieved": gathers / avg_s / 1e9 if avg_s > 0 else 0.0,
                     "peak": 78643.0, "unit": "GB/s", "frac": (gathers / avg_s / 1e9 / 78643.0) if avg_s > 0 else 0.0,
                     "note": "(nlist + L) x M fp32 table gathers per query against the ds_read_b32 peak (128 B/clk/CU)"}
    if coarse_s:
        r["coarse_prepass_ms"] = coarse_s * 1e3
        r["walk_kernel_ms"] = (avg_s - coarse_s) * 1e3
    r["note"] = ("algorithmic bytes = B x L x M candidate-row bytes + nlist x M centre bytes; coarse phase of the batch in "
                 "shard_coarse_quad_kernel (four queries per block, one 16-byte LDS read per centre lookup), then one block per query: "
                 "table loaded, the walk by one lane, the owned candidates (posting-order rows, next round prefetched)")
    return r


def readme_workload(args, torch, dev, arch):
    """configs[0]: the README example (N=10k, D=128, M=32, Ks=256, uniform random vectors; nlist = sqrt(N) = 100, topk = 3),
    ONE query per call through the host-pointer C ABI with a synchronisation per call -- the reference's own usage pattern --
    beside the reference on the host cores."""
    from rii_amd import RiiGpu
    from rii_amd import bench_data as bd
    rng = np.random.default_rng(0)
    N, D, M, Ks = 10_000, 128, 32, 256
    X = rng.random((N, D)).astype(np.float32)
    Q = rng.random((256, D)).astype(np.float32)
    cw = bd.train_pq(X[:5000], M, Ks, iters=5, seed=123, device=dev)
    codes = bd.encode_pq(X, cw, device=dev)
    eng = RiiGpu(cw, False, simd_arch=arch, device=dev.index)
    eng.add_codes(codes, False)
    eng.reconfigure(100, 5)
    topk, L = 3, 100
    E = np.array([], np.int64)
    out = {"config": "README example: N=10k D=128 M=32 Ks=256, nlist=100, ONE query per call (host pointers, one synchronisation "
                     "per call), topk=%d, L=%d" % (topk, L)}
    for name in ("linear", "ivf"):
        call = (lambda q: eng.query_ivf(q, topk, E, L)) if name == "ivf" else (lambda q: eng.query_linear(q, topk, E))
        for q in Q[:20]:
            call(q)
        ts, res = [], []
        for q in Q:
            t0 = time.perf_counter()
            r = call(q)
            ts.append(time.perf_counter() - t0)
            res.append(r[0])
        ts = np.array(ts) * 1e3
        o = {"p50_ms": float(np.percentile(ts, 50)), "p99_ms": float(np.percentile(ts, 99)), "value": 1e3 / float(np.mean(ts)),
             "unit": "queries/s"}
        if not args.no_cpu_baseline:
            cb, cpu_res = cpu_baseline(name, "README index, %s" % name, reference_factory(eng, cw, codes, arch, name == "ivf"),
                                       Q, topk, None, L, budget_s=1.0, thread_settings=[os.cpu_count() or 1, 8, 1])
            cb["ids_match_gpu"] = bool(all(list(res[b]) == list(cpu_res[b]) for b in range(len(cpu_res))))
            cb["queries_compared"] = len(cpu_res)
            o["cpu_baseline"] = cb
        out[name] = o
    out["note"] = ("latency-bound: linear = ONE launch (small_topk_kernel reads the query from and writes its rows to the engine's "
                   "pinned block, the host waits on a flag there); inverted index = ONE launch too (round 4: ivf_fused_kernel fetches the query from the pinned "
                   "block, replays a flagged query itself, rows and flag likewise); floor of an empty launch + flag on this box 7.5 us (tools/host_latency_probe.hip); no roofline "
                   "applies (the index is 320 KB)")
    return out


def deep_shard_workload(args, torch, dev, arch, barrier):
    """configs[4] per-GPU shape at a size that fits the default run: D=96, M=16, Ks=256 (Ds=6), --deep-shard codes of uniform
    random bytes (throughput only: no recall), batch 1024, top-1, one GPU = one shard."""
    from rii_amd import RiiGpu
    from rii_amd import bench_data as bd
    B, M, Ks, D = args.batch, 16, 256, 96
    n = args.deep_shard
    _, train, query = bd.sift_like(n_base=1, n_train=50_000, n_query=B, D=D, seed=99)
    cw = bd.train_pq(train, M, Ks, iters=5, seed=123, device=dev)
    codes = np.random.default_rng(1000).integers(0, 256, size=(n, M), dtype=np.uint8)
    eng = RiiGpu(cw, False, simd_arch=arch, device=dev.index)
    eng.add_codes(codes, False)
    for k in ("scan_mode", "scan_order", "scan_mx"):
        eng.set_option(k, getattr(args, k))
    q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
    out_ids = torch.empty((B, 1), dtype=torch.int64, device=dev)
    out_d = torch.empty((B, 1), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, out_ids.data_ptr(), out_d.data_ptr(), stream)

    steps = max(3, min(args.steps, 10))
    elapsed, dom, shares = measure(eng, step, steps, 2, barrier, torch.cuda.synchronize)
    plain = timed_loop(step, steps, barrier)
    k_ms, k_n = dom["scan"]
    avg_s = (k_ms / steps) * 1e-3
    filt = bool(args.scan_mode and B >= eng.get_option("fast_min_batch"))
    roof = roofline_scan(filter_kernel_name(args.scan_mx, M) if filt else "scan_kernel", B, n, M, Ks, avg_s, k_n, steps, filt,
                         1 if args.scan_mx or not filt else 2, workload_key("deep", args.scan_mode, args.scan_mx, M, n, B, 1))
    roof.update(shares)
    obj = {"config": "Deep1B-shaped shard: D=96 M=16 Ks=256, %d codes on one GPU, batch=%d, topk=1" % (n, B),
           "value": B * steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
           "uninstrumented_ms_per_step": plain / steps * 1e3, "uninstrumented_value": B * steps / plain,
           "kernel": roof["kernel"], "kernel_ms": avg_s * 1e3, "roofline": roof}
    res_ids = out_ids.cpu().numpy().copy()
    # one / two queries per call over the same shard (scan_kernel<1 / 2>: the exact fp32 scan as a pure HBM stream -- the code bytes
    # are read once per call, so bytes / kernel time IS the stream rate)
    few = {}
    for b in (1, 2):
        def few_step(b=b):
            eng.query_linear_dev(q.data_ptr(), b, 1, 0, 0, out_ids.data_ptr(), out_d.data_ptr(), stream)
        _e, fdom, _s = measure(eng, few_step, steps, 3, barrier, torch.cuda.synchronize)
        fplain = timed_loop(few_step, steps, barrier)
        fk = fdom["scan"][0] / steps
        few["B%d" % b] = {"ms_per_call": fplain / steps * 1e3, "kernel": "scan_kernel<%d>" % b, "kernel_ms": fk,
                          "stream_TBps": n * M / (fk * 1e-3) / 1e12, "hbm_frac_of_8TBps": n * M / (fk * 1e-3) / 8e12,
                          "ids_match_batch": bool((out_ids[:b].cpu().numpy() == res_ids[:b]).all())}
    obj["few_queries"] = few
    if not args.no_cpu_baseline:
        cb, cpu_res = cpu_baseline("linear", "full %d-code linear scan (M=16)" % n, reference_factory(eng, cw, codes, arch, False),
                                   query[:B], 1, None, 0, budget_s=3.0, thread_settings=[64, 16])
        cb["ids_match_gpu"], cb["queries_compared"] = ids_match(res_ids, None, cpu_res)
        obj["cpu_baseline"] = cb
    # the same shard through the DATABASE-SHARDED INVERTED INDEX (round 5): nlist = sqrt(N), L = N / nlist, one-rank communicator
    try:
        from rii_amd import dist as rd
        nl = int(np.round(np.sqrt(n)))
        obj["ivf_dbsharded"] = deep_ivf_on(eng, rd.get_comm(), args, torch, dev, barrier, q, n, n, 0, 1, nl, int(np.round(n / nl)), 1, steps,
                                           cw, codes, arch)
    except Exception as ex:                                      # noqa: BLE001
        obj["ivf_dbsharded"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    del eng
    return obj


def deep_structured_workload(args, torch, dev, arch, barrier):
    """configs[4] as a SEARCH (the 64 M-code leg is uniform random bytes with a synthetic partition -- bandwidth only): a
    structured Deep1B-shaped set (--deep-structured vectors, D = 96, unit-norm clustered descriptors), its own PQ codec (M = 16,
    Ks = 256), a REAL reconfigure(sqrt(N)) on the GPU (timed: row f1 of SURVEY 8 at scale -- PQk-means on the sample + the assignment
    of every code, src/rii.h:108-156), then linear and inverted-index search (L = N / nlist, examples/benchmark/run_sift1b.py:72-106)
    with recall@1 against the exact fp32 neighbours, beside the real reference on a few queries (it receives the GPU engine's centres
    and lists through its own pickle state, src/main.cpp:39-52)."""
    from rii_amd import RiiGpu
    from rii_amd import bench_data as bd
    B, M, Ks, D = args.batch, 16, 256, 96
    n = args.deep_structured
    t_all = time.perf_counter()
    base = bd.deep_like_torch(n, D, seed=77, device=dev)
    train = bd.deep_like_torch(100_000, D, seed=77, device=dev, stream=1)
    query = bd.deep_like_torch(B, D, seed=77, device=dev, stream=2)
    cw = bd.train_pq(train.cpu().numpy(), M, Ks, iters=8, seed=123, device=dev)
    codes_dev = bd.encode_pq_torch(base, cw)
    gt = bd.exact_nn_torch(base, query)
    del base, train
    codes = codes_dev.cpu().numpy()
    del codes_dev
    eng = RiiGpu(cw, False, simd_arch=arch, device=dev.index)
    eng.add_codes(codes, False)
    if getattr(args, "table_levels", 0):
        eng.set_option("generic_table_levels", args.table_levels)
    if getattr(args, "cand_cap", 0):
        eng.set_option("cand_cap", args.cand_cap)
    nlist = int(np.round(np.sqrt(n)))
    L = int(np.round(n / nlist))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.reconfigure(nlist, 5)
    eng.synchronize()
    t_reconf = time.perf_counter() - t0
    setup_s = time.perf_counter() - t_all
    q = query.contiguous()
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev)
    od = torch.empty((B, 1), dtype=torch.float32, device=dev)
    oc = torch.empty((B,), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    steps = max(3, min(args.steps, 10))
    obj = {"config": "structured Deep1B-shaped set: D=96 M=16 Ks=256, N=%d unit-norm clustered vectors, own PQ codec, nlist=%d "
                     "(reconfigure on the GPU), L=%d, batch=%d, topk=1" % (n, nlist, L, B),
           "reconfigure_s": t_reconf, "reconfigure": "rii_reconfigure(nlist=%d, iter=5): PQk-means on min(N, 100 nlist) sampled codes + "
                                                     "assignment of all %d codes (N x nlist x M = %.1e table gathers)" % (nlist, n, float(n) * nlist * M),
           "setup_s": setup_s}
    res = {}
    for name in ("ivf", "linear"):
        def step(name=name):
            if name == "ivf":
                eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), stream)
            else:
                eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), stream)
        elapsed, dom, shares = measure(eng, step, steps, 2, barrier, torch.cuda.synchronize)
        kn = ("ivf_fused" if dom["ivf_fused"][1] else "ivf_scan") if name == "ivf" else "scan"
        k_ms, k_n = dom[kn]
        ids = oi.cpu().numpy().copy()
        res[name] = ids
        row = {"value": B * steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3, "kernel_ms": k_ms / steps,
               "recall_at_1": float(bd.recall_at_r(ids, gt[:B], 1))}
        if name == "linear":
            row["filter_candidates_per_query"] = eng.get_option("cand_total") / float(B)
            row["table_levels"] = eng.get_option("generic_table_levels") or 255
        if name == "ivf":
            obj.update(row)
            obj["kernel"] = "ivf_fused_kernel"
            obj["counts_all_one"] = bool((oc.cpu().numpy() == 1).all())
        else:
            row["kernel"] = filter_kernel_name(args.scan_mx, M)
            obj["linear"] = row
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        ref, _arch, flav = O.load_reference()
        t0 = time.perf_counter()
        if ref is not None:
            ce = ref.RiiCpp.__new__(ref.RiiCpp)
            ce.__setstate__(eng.__getstate__())
            kind = "reference"
        else:
            ce = O.OracleRii(cw, False, simd_arch=arch)
            ce.add_codes(codes, False)
            ce.centers = np.array(eng.coarse_centers, np.uint8)
            ce._lists = eng.posting_lists
            kind = "port"
        t_state = time.perf_counter() - t0
        E = np.array([], np.int64)
        qh = query.cpu().numpy()
        nq = min(16, B)
        ce.query_ivf(qh[0], 1, E, L)
        t0 = time.perf_counter()
        cpu_ivf = [ce.query_ivf(qh[b], 1, E, L)[0] for b in range(nq)]
        dt = time.perf_counter() - t0
        cb = {"value": nq / dt, "unit": "queries/s", "cores": 1, "kind": kind,
              "sample": "inverted index nlist=%d L=%d over %d codes, %d queries, one per call; state handed over through the reference's "
                        "pickle hook in %.1f s%s" % (nlist, L, n, nq, t_state, (", build flavour " + flav) if flav else "")}
        cb["ids_match_gpu"] = bool(all(list(res["ivf"][b]) == list(cpu_ivf[b]) for b in range(nq)))
        cb["queries_compared"] = nq
        obj["cpu_baseline"] = cb
        nl = min(8, B)
        ce.query_linear(qh[0], 1, E)
        t0 = time.perf_counter()
        cpu_lin = [ce.query_linear(qh[b], 1, E)[0] for b in range(nl)]
        dt = time.perf_counter() - t0
        obj["linear"]["cpu_baseline"] = {"value": nl / dt, "unit": "queries/s", "cores": os.cpu_count() or 1, "kind": kind,
                                         "ids_match_gpu": bool(all(list(res["linear"][b]) == list(cpu_lin[b]) for b in range(nl))),
                                         "queries_compared": nl}
        del ce
    del eng
    return obj


def ref_harness_workload(args, torch, dev, arch, barrier, base_codes_src):
    """The reference's OWN SIFT1M harness configuration (examples/benchmark/ann_methods.py:19-34, run_sift1m.py:60-61): Rii(M=64,
    nlist=1000, L=5000), recall@1, one query per call -- measured both the harness's way (p50 of one-query host-pointer calls,
    run_sift1m.py:26-28) and as a batch of 1024, beside the real reference on the host cores."""
    from rii_amd import RiiGpu
    from rii_amd import bench_data as bd
    N, D, M, Ks, nlist, L, topk = args.n_base, 128, 64, 256, 1000, 5000, 1
    B = args.batch
    base, train, query, gt = base_codes_src()
    cw = bd.train_pq(train, M, Ks, iters=10, seed=123, device=dev)
    codes = bd.encode_pq(base, cw, device=dev)
    eng = RiiGpu(cw, False, simd_arch=arch, device=dev.index)
    eng.add_codes(codes, False)
    eng.reconfigure(nlist, 5)
    q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
    oi = torch.empty((B, topk), dtype=torch.int64, device=dev)
    od = torch.empty((B, topk), dtype=torch.float32, device=dev)
    oc = torch.empty((B,), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        eng.query_ivf_dev(q.data_ptr(), B, topk, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), stream)

    elapsed, dom, shares = measure(eng, step, args.steps, max(args.warmup, 2), barrier, torch.cuda.synchronize)
    plain = timed_loop(step, args.steps, barrier)
    res_ids, res_cnt = oi.cpu().numpy().copy(), oc.cpu().numpy().copy()
    kname = "ivf_fused" if dom["ivf_fused"][1] else "ivf_scan"
    k_ms, k_n = dom[kname]
    avg_s = (k_ms / max(args.steps, 1)) * 1e-3
    w = min(nlist, int(np.round(L * nlist / N)) + 3)
    key = "ref-harness/M=%d/N=%d/B=%d/nlist=%d/L=%d" % (M, N, B, nlist, L)
    roof = roofline_ivf(B, nlist, M, Ks, D // M, w, N // nlist, L, avg_s, k_n, args.steps,
                        profile_table("traffic.json").get(key, {}).get("hbm_bytes_per_launch"))
    # (the engine times its one-launch inverted-index kernels under one name; round 6's conflict-free gather is a kernel of its own)
    roof["kernel"] = "ivf_rot_kernel" if (kname == "ivf_fused" and eng.get_option("ivf_rot_launches") > 0) else kname + "_kernel"
    roof.update(shares)
    E = np.array([], np.int64)
    ts, one_ids = [], []
    for qq in query[:20]:
        eng.query_ivf(qq, topk, E, L)
    for qq in query[:256]:
        t0 = time.perf_counter()
        r = eng.query_ivf(qq, topk, E, L)
        ts.append(time.perf_counter() - t0)
        one_ids.append(r[0])
    ts = np.array(ts) * 1e3
    obj = {"config": "the reference's SIFT1M harness setting (examples/benchmark/ann_methods.py:19-34): SIFT1M-shaped, D=128 M=64 Ks=256, "
                     "N=%d, nlist=%d, L=%d, topk=1" % (N, nlist, L),
           "batch_%d" % B: {"value": B * args.steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / args.steps * 1e3,
                            "uninstrumented_ms_per_step": plain / args.steps * 1e3, "kernel": roof["kernel"], "kernel_ms": avg_s * 1e3},
           "one_query_per_call": {"p50_ms": float(np.percentile(ts, 50)), "p99_ms": float(np.percentile(ts, 99)), "value": 1e3 / float(ts.mean()),
                                  "unit": "queries/s", "what": "host pointers, one synchronisation per call (run_sift1m.py:26-28's loop)",
                                  "ids_match_batch": bool(all(list(one_ids[b]) == list(res_ids[b, :int(res_cnt[b])]) for b in range(len(one_ids))))},
           "recall_at_1": float(bd.recall_at_r(res_ids, gt[:B], 1)), "roofline": roof}
    if not args.no_cpu_baseline:
        cb, cpu_res = cpu_baseline("ivf", "inverted index M=64 nlist=%d L=%d" % (nlist, L), reference_factory(eng, cw, codes, arch, True),
                                   query[:B], topk, None, L, budget_s=3.0)
        cb["ids_match_gpu"], cb["queries_compared"] = ids_match(res_ids, res_cnt, cpu_res)
        obj["cpu_baseline"] = cb
    del eng
    return obj


def deep_ivf_on(eng, comm, args, torch, dev, barrier, q, n_shard, n_global, rank, world, nlist, L, topk, steps, cw, codes, arch):
    """The database-sharded inverted index over this rank's Deep1B-shaped shard, ONE rii_query_ivf_dbsharded_dev call per step (list
    lengths -> all-gather -> ivf_shard kernel -> pack -> all-gather -> merge -> finish).  Lists: modulo_lists() over the local ids,
    centres: nlist random codes, the same on every rank.  -> (object, rows)"""
    B, M = q.shape[0], eng.M
    centers = np.random.default_rng(4242).integers(0, 256, size=(nlist, M), dtype=np.uint8)
    off, ids = modulo_lists(n_shard, nlist)
    eng.set_posting_lists(centers, off, ids)
    oi = torch.empty((B, topk), dtype=torch.int64, device=dev)
    od = torch.empty((B, topk), dtype=torch.float32, device=dev)
    oc = torch.empty((B,), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        comm.query_ivf_dbsharded_dev(eng, rank * n_shard, n_global, q.data_ptr(), B, topk, 0, 0, 0, L, oi.data_ptr(), od.data_ptr(),
                                     oc.data_ptr(), 0, stream)

    elapsed, dom, shares = measure(eng, step, steps, 2, barrier, torch.cuda.synchronize, heat=8)
    plain = timed_loop(step, steps, barrier)
    k_ms, k_n = dom["ivf_shard"]
    c_ms, c_n = dom["shard_coarse"]
    avg_s = ((k_ms + c_ms) / max(steps, 1)) * 1e-3
    kname = "ivf_shard_any_kernel" if (L > 8192 or topk == 1) else "ivf_shard_kernel"
    key = "deep-ivf/M=%d/N=%d/B=%d/nlist=%d/L=%d" % (M, n_shard, B, nlist, L)
    roof = roofline_ivf_shard(("shard_coarse_quad_kernel + " if c_n else "") + kname, B, nlist, M, L, avg_s, k_n, steps,
                              profile_table("traffic.json").get(key, {}).get("hbm_bytes_per_launch"), (c_ms / max(steps, 1)) * 1e-3)
    roof.update(shares)
    obj = {"config": "Deep1B-shaped database-sharded inverted index: D=96 M=16 Ks=256, %d codes per GPU (%d in all), nlist=%d "
                     "(= sqrt(N)), L=%d (= N / nlist), batch=%d, topk=%d" % (n_shard, n_global, nlist, L, B, topk),
           "value": B * steps / elapsed, "unit": "queries/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps,
           "uninstrumented_ms_per_step": plain / steps * 1e3, "kernel": roof["kernel"], "kernel_ms": avg_s * 1e3, "roofline": roof,
           "lists": "synthetic partition (code i -> list i mod nlist; uniform random codes and centres): a real build's assignment "
                    "pass is N x nlist x M = %.1e table lookups" % (float(n_global) * nlist * M)}
    res_ids, res_cnt = oi.cpu().numpy().copy(), oc.cpu().numpy().copy()
    if world == 1 and not args.no_cpu_baseline:
        # the C oracle ("port"): the real reference takes posting lists through py::pickle's python lists only (src/main.cpp:39-52) --
        # 64 M python ints -- or through its own reconfigure (hours on this shape); the oracle is pinned to it bit for bit by tests/
        from oracle import oracle as O
        o = O.OracleRii(cw, False, simd_arch=arch)
        o.codes = codes
        o.set_csr(centers, off, ids)
        E = np.array([], np.int64)
        qh = q.cpu().numpy()
        nq = min(16, B)
        o.query_ivf(qh[0], topk, E, L)
        t0 = time.perf_counter()
        cpu_res = [o.query_ivf(qh[b], topk, E, L)[0] for b in range(nq)]
        dt = time.perf_counter() - t0
        cb = {"value": nq / dt, "unit": "queries/s", "ms_per_query": dt / nq * 1e3, "cores": 1, "kind": "port",
              "sample": "one query per call, inverted index nlist=%d L=%d over %d codes, %d queries (QueryIvf is single-threaded by "
                        "design, src/rii.h:244-326)" % (nlist, L, n_shard, nq)}
        cb["ids_match_gpu"], cb["queries_compared"] = ids_match(res_ids, res_cnt, cpu_res)
        obj["cpu_baseline"] = cb
    return obj


def main_deep_ivf(args, world, rank, local, dev, arch):
    """`--workload deep-ivf`: BASELINE configs[4]'s shards searched the way the reference searches a billion vectors
    (examples/benchmark/run_sift1b.py:105-106: nlist = sqrt(N), L = N / nlist) -- the database-sharded inverted index through
    rii_query_ivf_dbsharded_dev, --n-base codes PER GPU."""
    import torch
    import torch.distributed as dist
    from rii_amd import RiiGpu
    from rii_amd import bench_data as bd
    from rii_amd import dist as rd
    B, M, Ks, D = args.batch, 16, 256, 96
    n_shard = args.n_base
    n_global = n_shard * world
    nlist = args.nlist if args.nlist > 0 else int(np.round(np.sqrt(n_global)))
    L = args.L if args.L > 0 else int(np.round(n_global / nlist))
    _, train, query = bd.sift_like(n_base=1, n_train=50_000, n_query=B, D=D, seed=99)
    cw = bd.train_pq(train, M, Ks, iters=5, seed=123, device=dev)
    codes = np.random.default_rng(1000 + rank).integers(0, 256, size=(n_shard, M), dtype=np.uint8)
    eng = RiiGpu(cw, False, simd_arch=arch, device=local)
    if args.shard_dbg_stop:
        eng.set_option("shard_dbg_stop", args.shard_dbg_stop)
    eng.add_codes(codes, False)
    q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
    use_dist = dist.is_initialized()
    comm = rd.get_comm()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    obj = deep_ivf_on(eng, comm, args, torch, dev, barrier, q, n_shard, n_global, rank, world, nlist, L, args.topk, args.steps, cw, codes, arch)
    elapsed_ms = obj["ms_per_step"]
    if use_dist:
        t = torch.tensor([elapsed_ms], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_ms = float(t.item())
    if rank == 0:
        line = {"metric": "queries/sec", "value": B / elapsed_ms * 1e3, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": obj["config"], "global_batch": B,
                           "parallelism": "database-sharded x%d: ONE rii_query_ivf_dbsharded_dev call per step (two RCCL all-gathers + device "
                                          "merge in the timed region; top-1 never synchronises with the host)" % world},
                "recall_at_1": None, "roofline": obj["roofline"], "uninstrumented_ms_per_step": obj["uninstrumented_ms_per_step"],
                "lists": obj["lists"]}
        if "cpu_baseline" in obj:
            line["cpu_baseline"] = obj["cpu_baseline"]
        emit(line, args)
    rd.close_comms()
    if use_dist:
        dist.destroy_process_group()


# --------------------------------------------------------------------------------------------------------------------
def main_deep(args, world, rank, local, dev, arch):
    """Deep1B-shaped database sharding (BASELINE configs[4]): D=96, M=16, Ks=256; every rank holds --n-base codes
    (uniform random bytes: throughput only, so no recall), all ranks answer the SAME batch on their shard through
    rii_amd.dist.DbShardedIndex: engine -> record -> RCCL all-gather -> device merge (per-rank id offsets added by the merge
    kernel), no host synchronisation in the step."""
    import torch
    import torch.distributed as dist
    from rii_amd import RiiGpu
    from rii_amd import bench_data as bd
    from rii_amd import dist as rd
    B, M, Ks, D = args.batch, 16, 256, 96
    n_shard = args.n_base
    _, train, query = bd.sift_like(n_base=1, n_train=50_000, n_query=B, D=D, seed=99)
    cw = bd.train_pq(train, M, Ks, iters=5, seed=123, device=dev)
    rng = np.random.default_rng(1000 + rank)
    codes = rng.integers(0, 256, size=(n_shard, M), dtype=np.uint8)
    eng = RiiGpu(cw, False, simd_arch=arch, device=local)
    eng.add_codes(codes, False)
    eng.set_option("scan_mode", args.scan_mode)
    eng.set_option("scan_order", args.scan_order)
    eng.set_option("scan_mx", args.scan_mx)
    del codes
    topk = args.topk
    q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
    use_dist = dist.is_initialized()
    idx = rd.DbShardedIndex(eng, rank * n_shard, (rank + 1) * n_shard)
    idx.all_starts()
    merged = [None]

    def step():
        merged[0] = idx.query_linear_batch(q, topk)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # (the step holds a collective: pre-heat by step count, the same on every rank)
    elapsed, dom, shares = measure(eng, step, args.steps, args.warmup, barrier, torch.cuda.synchronize,
                                   heat=max(8, min(200, int(0.15 / max(1e-4, 2e-10 * n_shard * B / 1024.0)))))
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms, k_n = dom["scan"]
    if rank == 0:
        avg_s = (k_ms / max(args.steps, 1)) * 1e-3
        filt = bool(args.scan_mode and (topk > 1 or B >= eng.get_option("fast_min_batch")))
        key = workload_key("deep", args.scan_mode, args.scan_mx, M, n_shard, B, topk)
        if B == 1:
            roof = roofline_hbm("scan_kernel", n_shard * M, avg_s, k_n, args.steps,
                                profile_table("traffic.json").get(key, {}).get("hbm_bytes_per_launch"))
        else:
            roof = roofline_scan(filter_kernel_name(args.scan_mx, M) if filt else "scan_kernel", B, n_shard, M, Ks, avg_s, k_n,
                                 args.steps, filt, 1 if args.scan_mx or not filt else 2, key)
        roof.update(shares)
        print(json.dumps({
            "metric": "queries/sec", "value": B * args.steps / elapsed, "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Deep1B-shaped linear ADC scan, D=96 M=16 Ks=256, %d codes per GPU (database sharded, "
                                   "%d codes total), batch=%d, topk=%d" % (n_shard, n_shard * world, B, topk),
                       "global_batch": B,
                       "parallelism": "database-sharded x%d through rii_amd.dist.DbShardedIndex: RCCL all-gather + device (dist,id) "
                                      "merge in the timed region, no host synchronisation per step" % world,
                       "scan_mode": "byte-table filter + exact fp32 re-rank" if args.scan_mode else "exact fp32 scan"},
            "recall_at_1": None, "roofline": roof}))
    if use_dist:
        dist.destroy_process_group()


def main():
    global PREHEAT_S
    t_start = time.perf_counter()
    wall = {}
    args = parse()
    PREHEAT_S = max(0.0, args.preheat)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    import torch
    import torch.distributed as dist
    from rii_amd import RiiGpu, host_simd_arch
    from rii_amd import bench_data as bd
    from rii_amd import dist as rd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = "WORLD_SIZE" in os.environ           # launched by torch.distributed.run (any world size, incl. 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("RII_BENCH_BACKEND", "nccl")     # "gloo": debugging the N>1 logic on a box with one GPU
        if backend != "nccl":
            local = int(os.environ.get("RII_BENCH_DEVICE", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (run `python bench.py --gpus N` directly, or under "
                         "`python -m torch.distributed.run --nproc-per-node N`)" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, M, Ks, D = args.batch, args.M, 256, 128
    arch = host_simd_arch()
    if args.workload == "deep":
        return main_deep(args, world, rank, local, dev, arch)
    if args.workload == "deep-ivf":
        return main_deep_ivf(args, world, rank, local, dev, arch)

    # ---------------- inputs (synthetic, seeded): rank 0 builds, everyone receives ----------------
    N = args.n_base
    nq = max(B * world, 1024)
    if rank == 0:
        base, train, query = bd.sift_like(n_base=N, n_train=100_000, n_query=max(10_000, nq), D=D)
        cw = bd.train_pq(train, M, Ks, iters=10, seed=123, device=dev)
        codes = bd.encode_pq(base, cw, device=dev)
        gt = bd.exact_nn(base, query[:nq], device=dev)
        t_cw = torch.from_numpy(cw).to(dev)
        t_codes = torch.from_numpy(codes).to(dev)
        t_q = torch.from_numpy(np.ascontiguousarray(query[:nq])).to(dev)
        t_gt = torch.from_numpy(gt).to(dev)
        # (the `others.ref_harness` leg re-encodes the same vectors with M = 64: kept for the default invocation only)
        keep_vectors = (base, train, query) if (world == 1 and not args.no_others and args.workload == "linear" and args.topk == 1) else None
        del base, train
    else:
        t_cw = torch.empty((M, Ks, D // M), dtype=torch.float32, device=dev)
        t_codes = torch.empty((N, M), dtype=torch.uint8, device=dev)
        t_q = torch.empty((nq, D), dtype=torch.float32, device=dev)
        t_gt = torch.empty((nq,), dtype=torch.int64, device=dev)
    host_coll = use_dist and dist.get_backend() != "nccl"         # gloo: run the collectives on host copies

    if use_dist:
        for t in (t_cw, t_codes, t_q, t_gt):
            if host_coll:
                c = t.cpu()
                dist.broadcast(c, 0)
                t.copy_(c)
            else:
                dist.broadcast(t, 0)
    cw = t_cw.cpu().numpy()
    codes = t_codes.cpu().numpy()
    del t_codes
    my_q = t_q[rank * B:(rank + 1) * B].contiguous()
    my_gt = t_gt[rank * B:(rank + 1) * B].cpu().numpy()

    # ---------------- engine (one per GPU, index replicated) ----------------
    eng = RiiGpu(cw, False, simd_arch=arch, device=local)
    eng.add_codes(codes, False)
    eng.set_option("lut_mode", args.lut_mode)
    eng.set_option("scan_mode", args.scan_mode)
    eng.set_option("scan_order", args.scan_order)
    eng.set_option("scan_mx", args.scan_mx)
    topk = args.topk
    ivf = args.workload in ("ivf", "subset-ivf")
    S, L, d_tids, h_tids = 0, 0, 0, None
    nlist_main = args.nlist if args.nlist > 0 else 1024
    if ivf:
        eng.reconfigure(nlist_main, 5)
        L = args.L if args.L > 0 else int(np.round(N / nlist_main))
    if args.workload in ("subset", "subset-ivf"):
        h_tids = np.sort(np.random.default_rng(7).choice(N, min(100_000, N), replace=False)).astype(np.int64)
        tids = torch.from_numpy(h_tids).to(dev)
        S, d_tids = tids.numel(), tids.data_ptr()
    out_ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
    out_d = torch.empty((B, topk), dtype=torch.float32, device=dev)
    out_cnt = torch.empty((B,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)      # engine kernels and RCCL calls are ordered on ONE torch stream
    torch.cuda.set_stream(side)
    stream = side.cuda_stream

    def run(qt, oi=None, od=None, oc=None):
        oi, od, oc = oi if oi is not None else out_ids, od if od is not None else out_d, oc if oc is not None else out_cnt
        if ivf:
            eng.query_ivf_dev(qt.data_ptr(), qt.shape[0], topk, d_tids, S, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), stream)
        else:
            eng.query_linear_dev(qt.data_ptr(), qt.shape[0], topk, d_tids, S, oi.data_ptr(), od.data_ptr(), stream)

    # `value` (round 4): SURVEY 8d's metric counts the device->host of the results, so the step that is timed hands the engine HOST
    # rows -- two pinned buffers in rotation; the last kernel of the step (re-rank / inverted-index kernel) writes its rows straight
    # into them over PCIe, no D2H copy is enqueued -- and the closing barrier of the timed region is what makes the last steps' rows
    # visible.  Queries stay resident in HBM.  `device_resident` below is the same loop with the rows left in HBM, `sync_call` one
    # rii_query_*_dev_to_host call per step (rows in the caller's arrays when each call returns).
    host_rows = [(torch.empty((B, topk), dtype=torch.int64).pin_memory(), torch.empty((B, topk), dtype=torch.float32).pin_memory(),
                  torch.empty((B,), dtype=torch.int64).pin_memory()) for _ in range(2)]
    it_h = [0]

    def step():
        # query sharding: every rank owns the results of its own queries, no exchange
        oi, od, oc = host_rows[it_h[0] & 1]
        it_h[0] += 1
        run(my_q, oi, od, oc)

    def step_dev():
        run(my_q)

    gathered = [None]

    def step_gather():
        run(my_q)
        gathered[0] = rd.allgather_query_shards(out_ids, out_d)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(*vals):
        if not use_dist:
            return vals
        t = torch.tensor(list(vals), dtype=torch.float64, device="cpu" if host_coll else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    if args.latency:
        return main_latency(args, eng, t_q, run, ivf, topk, h_tids, L, rank, world, dev)

    # first exactly what the contract's wording gives from an idle GPU (W warm-up + K timed steps, clocks still ramping) ...
    for _ in range(args.warmup):
        step()
    wall["setup_s"] = time.perf_counter() - t_start          # imports, synthetic data, codebooks, encoding, ground truth, index upload
    cold = timed_loop(step, args.steps, barrier)
    # ... then the steady state: pre-heat, W warm-up steps, K timed steps (this is `value`)
    elapsed, dom, extra = measure(eng, step, args.steps, args.warmup, barrier, torch.cuda.synchronize)
    last = host_rows[(it_h[0] - 1) & 1]
    res_ids = last[0].numpy().copy()                 # (host memory: the rows the timed steps delivered)
    res_cnt = last[2].numpy().copy() if ivf else None
    rows_both_buffers_equal = bool(np.array_equal(host_rows[0][0].numpy(), host_rows[1][0].numpy()))
    # the same K steps with no timing event at all (twice): what the events attached to the dominant kernel's dispatches cost,
    # and how stable a K-step sample is
    plain = [timed_loop(step, args.steps, barrier) for _ in range(2)]
    # ... with the rows left in HBM (round 3's `value`), and one synchronous device-queries -> host-rows call per step
    dev_res = timed_loop(step_dev, args.steps, barrier)
    assert np.array_equal(out_ids.cpu().numpy(), res_ids), "host-delivered rows differ from the device-resident ones"
    h_i, h_d, h_c = np.empty((B, topk), np.int64), np.empty((B, topk), np.float32), np.empty(B, np.int64)

    def step_sync():
        if ivf:
            eng.query_ivf_dev_to_host(my_q.data_ptr(), B, topk, d_tids, S, L, h_i, h_d, h_c, stream)
        else:
            eng.query_linear_dev_to_host(my_q.data_ptr(), B, topk, d_tids, S, h_i, h_d, stream)

    for _ in range(max(args.warmup, 1)):
        step_sync()
    sync_call = timed_loop(step_sync, args.steps, barrier)
    assert np.array_equal(h_i, res_ids)
    elapsed_g = None
    if use_dist:
        preheat_count(step_gather, torch.cuda.synchronize, 100)
        for _ in range(args.warmup):
            step_gather()
        elapsed_g = timed_loop(step_gather, args.steps, barrier)
        elapsed, elapsed_g, plain[0], plain[1], cold, dev_res, sync_call = max_over_ranks(elapsed, elapsed_g, plain[0], plain[1], cold, dev_res, sync_call)
        allq = gathered[0][0]                        # the gathered batch really is every rank's rows, in rank order
        assert allq.shape[0] == B * world and torch.equal(allq[rank * B:(rank + 1) * B].to(out_ids.device), out_ids)

    # ---------------- strong scaling: BASELINE's "batch=1024 at 1/2/4/8 GPU" -- ONE global batch split over the ranks ----------------
    strong = None
    if use_dist and not args.no_strong:
        strong = {}
        Qg = t_q[:B].contiguous()                    # the same global batch on every rank
        Qs = Qg.cpu().numpy() if host_coll else Qg   # (gloo: host engines' surface)
        tids_np = h_tids if (host_coll or not S) else tids
        # a failure of either form is reported in its object instead of taking the whole line down (the weak-scaling `value`
        # above is already measured); the same exception on every rank keeps the ranks in step
        try:
            qidx = rd.QueryShardedIndex(eng)
            res_q = [None]
            out_q = None if host_coll else (torch.empty((B, topk), dtype=torch.int64, device=dev), torch.empty((B, topk), dtype=torch.float32, device=dev))

            def step_strong_q():
                if ivf:
                    res_q[0] = qidx.query_ivf_batch(Qs, topk, tids_np, L)
                else:
                    res_q[0] = qidx.query_linear_batch(Qs, topk, tids_np, out=out_q)

            preheat_count(step_strong_q, torch.cuda.synchronize, 100)
            for _ in range(max(args.warmup, 1)):
                step_strong_q()
            e_q, = max_over_ranks(timed_loop(step_strong_q, args.steps, barrier))
            run(Qg)                                      # every row equals the single-engine answer for that row
            torch.cuda.synchronize()
            ok_q = bool(torch.equal(torch.as_tensor(res_q[0][0]).to(dev), out_ids))
            strong["query_sharded"] = {"value": B * args.steps / e_q, "unit": "queries/s", "ms_per_step": e_q / args.steps * 1e3,
                                       "global_batch": B, "rows_per_rank": [rd.shard_range(B, r, world)[1] - rd.shard_range(B, r, world)[0]
                                                                              for r in range(world)],
                                       "results_match_single_engine": ok_q,
                                       "what": "index replicated, rank r answers its slice of the global batch, ONE all-gather of the packed result rows "
                                               "inside the timed region -- engine kernels, ncclAllGather and the unpack kernel enqueued by ONE "
                                               "C-ABI call (rii_query_linear_qsharded_dev; QueryShardedIndex is its thin caller)"}
        except Exception as ex:                              # noqa: BLE001
            strong["query_sharded"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if not ivf and S == 0:
            try:
                s0, s1 = rd.shard_range(N, rank, world)
                eng_s = RiiGpu(cw, False, simd_arch=arch, device=local)
                eng_s.add_codes(codes[s0:s1], False)
                for k in ("scan_mode", "scan_order", "scan_mx"):
                    eng_s.set_option(k, getattr(args, k))
                didx = rd.DbShardedIndex(eng_s, s0, s1)
                didx.all_starts()
                res_d = [None]
                out_dd = None if host_coll else (torch.empty((B, topk), dtype=torch.int64, device=dev), torch.empty((B, topk), dtype=torch.float32, device=dev))

                def step_strong_d():
                    res_d[0] = didx.query_linear_batch(Qs, topk, out=out_dd)

                preheat_count(step_strong_d, torch.cuda.synchronize, 100)
                for _ in range(max(args.warmup, 1)):
                    step_strong_d()
                e_d, = max_over_ranks(timed_loop(step_strong_d, args.steps, barrier))
                ok_d = bool(torch.equal(torch.as_tensor(res_d[0][0]).to(dev), out_ids) and torch.equal(torch.as_tensor(res_d[0][1]).to(dev), out_d))
                strong["db_sharded"] = {"value": B * args.steps / e_d, "unit": "queries/s", "ms_per_step": e_d / args.steps * 1e3,
                                        "global_batch": B, "codes_per_rank": s1 - s0, "results_match_single_engine": ok_d,
                                        "what": "codes split into contiguous id ranges, every rank answers the whole batch on its shard, ONE all-gather "
                                                "+ device (dist, id) merge inside the timed region, all enqueued by ONE C-ABI call "
                                                "(rii_query_linear_dbsharded_dev; no host synchronisation for top-1)"}
                del didx, eng_s
            except Exception as ex:                          # noqa: BLE001
                strong["db_sharded"] = {"error": "%s: %s" % (type(ex).__name__, ex)}

        run(my_q)
        torch.cuda.synchronize()

    kernel = "scan"
    if ivf:
        kernel = "ivf_fused" if eng.get_option("ivf_fused") else "ivf_scan"
    k_ms, k_n = dom[kernel]
    recall = bd.recall_at_r(res_ids, my_gt, 1)
    if use_dist:
        r = torch.tensor([recall], dtype=torch.float64, device="cpu" if host_coll else dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        recall = float(r.item()) / world

    # the same step through the host-pointer C ABI: queries from host memory, (ids, dists) back to host memory
    host = None
    if rank == 0 and not args.no_host_call:
        hq = my_q.cpu().numpy()
        call = (lambda: eng.query_ivf_batch(hq, topk, h_tids, L)) if ivf else (lambda: eng.query_linear_batch(hq, topk, h_tids))
        for _ in range(max(args.warmup, 1)):
            call()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            call()
        he = time.perf_counter() - t0
        host = {"ms_per_step": he / args.steps * 1e3, "value": B * args.steps / he, "unit": "queries/s",
                "what": "rii_query_%s with host pointers: H2D of %d B of queries, the step, D2H of %d B of results, "
                        "one synchronisation per call" % ("ivf" if ivf else "linear", hq.nbytes, B * topk * 12 + (8 * B if ivf else 0))}
    # A different batch every step (the timed loop above re-submits one batch; the engine caches nothing between calls --
    # tables are rebuilt per call -- so this is the same number, measured rather than argued)
    fresh = None
    if world == 1 and not args.no_fresh:
        more = bd.more_queries(7 * B, D=D)                 # same distribution as the timed batch (same cluster means)
        pool = [my_q] + [torch.from_numpy(more[i * B:(i + 1) * B]).to(dev) for i in range(7)]
        torch.cuda.synchronize()
        it_f = [0]

        def step_fresh():
            run(pool[it_f[0] % len(pool)])
            it_f[0] += 1

        preheat(step_fresh, torch.cuda.synchronize, 0.05)
        for _ in range(max(args.warmup, 1)):
            step_fresh()
        fe = timed_loop(step_fresh, args.steps, barrier)
        fresh = {"value": B * args.steps / fe, "unit": "queries/s", "ms_per_step": fe / args.steps * 1e3,
                 "what": "%d distinct query batches in rotation, one stream" % len(pool)}
        run(my_q)
        torch.cuda.synchronize()
        del pool

    # The same K steps issued alternately on two HIP streams: the engine keeps one scratch lane per stream (engine.hip:
    # ScratchSet), so the latency-bound phases of one step (table build, re-rank, launch gaps) overlap the other step's
    # scan.  Reported beside `value`; `value` and the roofline stay the one-stream numbers.  The second stream answers the
    # batch in reverse order; both are compared with one-stream results.
    pipe = None
    if world == 1 and not args.no_pipelined:
        reps = []
        for i in range(2):
            st = torch.cuda.Stream(device=dev)
            reps.append({"eng": eng, "st": st, "q": (my_q if i == 0 else my_q.flip(0)).contiguous(),
                         "ids": torch.empty_like(out_ids), "d": torch.empty_like(out_d), "cnt": torch.empty_like(out_cnt)})
        torch.cuda.synchronize()

        def run_rep(r):
            q_, s_ = r["q"], r["st"].cuda_stream
            if ivf:
                r["eng"].query_ivf_dev(q_.data_ptr(), B, topk, d_tids, S, L, r["ids"].data_ptr(), r["d"].data_ptr(),
                                       r["cnt"].data_ptr(), s_)
            else:
                r["eng"].query_linear_dev(q_.data_ptr(), B, topk, d_tids, S, r["ids"].data_ptr(), r["d"].data_ptr(), s_)

        it = [0]

        def step_pipe():
            run_rep(reps[it[0] & 1])
            it[0] += 1

        preheat(step_pipe, torch.cuda.synchronize, 0.05)
        for _ in range(2 * max(args.warmup, 1)):
            step_pipe()
        pe = timed_loop(step_pipe, args.steps, barrier)
        same = True
        for r in reps:
            run(r["q"])
            torch.cuda.synchronize()
            same = same and torch.equal(out_ids, r["ids"]) and torch.equal(out_d, r["d"])
        run(my_q)
        torch.cuda.synchronize()
        pipe = {"value": B * args.steps / pe, "unit": "queries/s", "ms_per_step": pe / args.steps * 1e3,
                "results_match_one_stream": bool(same),
                "what": "the same engine answering alternate batches on two HIP streams (one scratch lane per stream)"}
        del reps
    if use_dist:
        barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = B * world * args.steps / elapsed
        n_scanned = S if args.workload == "subset" else N
        avg_s = (k_ms / max(args.steps, 1)) * 1e-3          # top-k runs the scan kernel twice per step: charged together
        key = workload_key(args.workload, args.scan_mode, args.scan_mx, M, n_scanned, B, topk)
        if ivf:
            w = min(nlist_main, int(np.round(L * nlist_main / (S if S else N))) + 3)
            roof = roofline_ivf(B, nlist_main, M, Ks, D // M, w, N // nlist_main, L, avg_s, k_n, args.steps,
                                profile_table("traffic.json").get(key, {}).get("hbm_bytes_per_launch"))
        else:
            filt = bool(args.scan_mode and (topk > 1 or B >= eng.get_option("fast_min_batch")))
            roof = roofline_scan(filter_kernel_name(args.scan_mx, M) if filt else "scan_kernel", B, n_scanned, M, Ks, avg_s, k_n,
                                 args.steps, filt, 1 if args.scan_mx or not filt else 2, key)
        roof.update(extra)
        if world == 1 and not args.no_live_counters:
            shape = ["--workload", args.workload, "--batch", str(B), "--topk", str(topk), "--scan-mode", str(args.scan_mode),
                     "--scan-mx", str(args.scan_mx), "--scan-order", str(args.scan_order), "--lut-mode", args.lut_mode,
                     "--n-base", str(N), "--M", str(M), "--nlist", str(args.nlist), "--L", str(args.L)]
            t_lc = time.perf_counter()
            lc = live_counters(roof["kernel"], shape)
            wall["live_counters_s"] = time.perf_counter() - t_lc
            if "FETCH_SIZE" in lc and "WRITE_SIZE" in lc:
                # counters are in KB; FETCH_SIZE reports half of a wide coalesced stream on gfx950 (MI355X guide): doubled
                traffic = int((2.0 * lc["FETCH_SIZE"] + lc["WRITE_SIZE"]) * 1024.0)
                roof["traffic_from_profiles"] = roof.get("traffic")
                roof["traffic"] = traffic
                roof["traffic_source"] = ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a 3-step copy of this command, run by "
                                          "bench.py itself (%d dispatches of the kernel); (2 x FETCH_SIZE + WRITE_SIZE) x 1024" % lc.get("_dispatches", 0))
                if isinstance(roof.get("hbm"), dict) and avg_s > 0:
                    roof["hbm"]["traffic_bytes"] = traffic
                    roof["hbm"]["achieved"] = traffic / avg_s / 1e9
                    roof["hbm"]["frac"] = roof["hbm"]["achieved"] / HBM_PEAK_GBPS
            if lc.get("SQ_LDS_IDX_ACTIVE") and lc.get("SQ_INSTS_LDS"):
                roof["counters_live"] = {"lds_conflict_frac": lc.get("SQ_LDS_BANK_CONFLICT", 0.0) / lc["SQ_LDS_IDX_ACTIVE"],
                                         "lds_cycles_per_read": lc["SQ_LDS_IDX_ACTIVE"] / lc["SQ_INSTS_LDS"],
                                         "lds_insts": lc["SQ_INSTS_LDS"], "valu_insts": lc.get("SQ_INSTS_VALU"),
                                         "wave_wait_frac": (lc.get("SQ_WAIT_ANY", 0.0) / lc["SQ_WAVE_CYCLES"]) if lc.get("SQ_WAVE_CYCLES") else None,
                                         "source": "live rocprofv3 --pmc pass of this run"}
                if lc.get("GRBM_GUI_ACTIVE") and avg_s > 0:
                    # shader cycles of one launch (the counter sums the 8 XCDs) over the launch time of the timed region: the clock the
                    # chip held under this kernel -- `frac` is quoted against the NOMINAL 2.4 GHz, boxes of the pool hold 2.0 - 2.2
                    cyc = lc["GRBM_GUI_ACTIVE"] / 8.0
                    ghz = cyc / avg_s / 1e9
                    roof["counters_live"].update({"gpu_cycles_per_launch": cyc, "shader_clock_ghz": ghz,
                                                  "lds_busy": lc["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc,
                                                  "frac_at_held_clock": roof["frac"] * 2.4 / ghz if ghz > 0 else None})
        line = {
            "metric": "queries/sec", "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SIFT1M-shaped %s search, D=128 M=%d Ks=256, N=%d, batch=%d per GPU, topk=%d%s%s"
                                   % (args.workload, M, N, B, topk, (", nlist=%d L=%d" % (nlist_main, L)) if L else "",
                                      (", |target_ids|=%d" % S) if S else ""),
                       "global_batch": B * world,
                       "parallelism": "query-sharded x%d, index replicated; value: no exchange step (results stay with the "
                                      "owning rank), with_gather: RCCL all-gather of the result rows in the timed region; "
                                      "strong: ONE global batch of %d split over the ranks" % (world, B),
                       "lut_mode": args.lut_mode, "simd_order": arch,
                       "scan_mode": "8-bit filter + exact fp32 re-rank" if args.scan_mode else "exact fp32 scan"},
            "recall_at_1": recall,
            "results": {"delivery": "host: the last kernel of every timed step writes its rows (ids int64, dists f32%s) straight into pinned host "
                                    "memory (two buffers in rotation, %d B per step over PCIe; no D2H copy, no per-step synchronisation); queries "
                                    "resident in HBM" % (", counts int64" if ivf else "", B * topk * 12 + (8 * B if ivf else 0)),
                        "rows_checked": "host rows == device-resident rows == reference rows (cpu_baseline.ids_match_gpu); both host "
                                        "buffers hold the batch's rows: %s" % rows_both_buffers_equal,
                        "device_resident": {"ms_per_step": dev_res / args.steps * 1e3, "value": B * world * args.steps / dev_res,
                                            "unit": "queries/s", "what": "the same K steps with the rows left in HBM (round 3's `value`)"},
                        "sync_call": {"ms_per_step": sync_call / args.steps * 1e3, "value": B * world * args.steps / sync_call, "unit": "queries/s",
                                      "what": "one rii_query_%s_dev_to_host call per step: queries in HBM, rows in the caller's host arrays when "
                                              "EACH call returns (one host wait per step: nothing of step i+1 is enqueued before step i "
                                              "has landed)" % ("ivf" if ivf else "linear")}},
            "roofline": roof,
            "preheat": {"seconds": PREHEAT_S, "what": "untimed steps before the W warm-up steps: the shader clock ramps over the first ~30 ms "
                                                        "of sustained load (profiles/r03_clock_ramp.json); `value` is the steady state",
                        "cold_start_ms_per_step": cold / args.steps * 1e3,
                        "cold_start_value": B * world * args.steps / cold,
                        "cold_start_what": "W warm-up + K timed steps issued right after the index build, before any pre-heat"},
            "uninstrumented": {"ms_per_step": [p_ / args.steps * 1e3 for p_ in plain],
                               "value": B * world * args.steps / min(plain), "unit": "queries/s",
                               "what": "the same K steps twice more with no timing event in the stream (`value`'s loop carries two "
                                       "events on the dominant kernel's dispatch, which the roofline needs)"},
        }
        if elapsed_g is not None:
            line["with_gather"] = {"ms_per_step": elapsed_g / args.steps * 1e3, "value": B * world * args.steps / elapsed_g,
                                   "unit": "queries/s", "backend": dist.get_backend(),
                                   "collective": "all_gather of %d B per rank (ids int64 + dists f32), device tensors" % (B * topk * 12),
                                   "collective_share": max(0.0, 1.0 - elapsed / elapsed_g)}
        if strong:
            line["strong"] = strong
            # BASELINE's metric is "batch = 1024 at 1 / 2 / 4 / 8 GPU": the strong-scaling figures at the top level, beside the weak
            # `value` the contract asks for (the first real SCALE run must be readable without digging)
            line["strong_query_sharded_value"] = strong.get("query_sharded", {}).get("value")
            line["strong_db_sharded_value"] = strong.get("db_sharded", {}).get("value")
        if use_dist:
            line["rccl_ranks"] = world if dist.get_backend() == "nccl" else 0
        if host is not None:
            line["host_call"] = host
        if fresh is not None:
            line["fresh_queries"] = fresh
        if pipe is not None:
            line["pipelined"] = pipe
        if world == 1 and not args.no_cpu_baseline:
            what = {"linear": "full %d-code linear scan" % N, "subset": "linear scan of %d target ids" % S,
                    "ivf": "inverted index nlist=%d L=%d" % (nlist_main, L),
                    "subset-ivf": "inverted index nlist=%d L=%d over %d target ids" % (nlist_main, L, S)}
            t_cb = time.perf_counter()
            try:
                cb, cpu_res = cpu_baseline("ivf" if ivf else "linear", what[args.workload], reference_factory(eng, cw, codes, arch, ivf),
                                           my_q.cpu().numpy(), topk, h_tids, L)
                cb["ids_match_gpu"], cb["queries_compared"] = ids_match(res_ids, res_cnt, cpu_res)
            except Exception as ex:                              # noqa: BLE001 -- the measured line above must still be printed
                cb = {"error": "%s: %s" % (type(ex).__name__, ex)}
            line["cpu_baseline"] = cb
            wall["cpu_baseline_s"] = time.perf_counter() - t_cb
        # every other BASELINE config in the same process (default invocation only: the SIFT-shaped legs reuse this index)
        if world == 1 and not use_dist and not args.no_others and args.workload == "linear" and topk == 1:
            others = {}
            t_oth = time.perf_counter()
            def guarded(fn, *a):                                 # one failing leg must not take the line (or the other legs) down
                try:
                    return fn(*a)
                except Exception as ex:                              # noqa: BLE001
                    return {"error": "%s: %s" % (type(ex).__name__, ex)}

            for name in ("subset", "ivf", "subset_ivf"):            # (reconfigure happens once, before the two ivf legs)
                others[name] = guarded(sift_workload, name, eng, args, torch, dev, stream, my_q, cw, codes, barrier, arch, N, M, Ks,
                                       D // M, B, topk)
            others["sharded_world1"] = guarded(sharded_world1_workload, eng, args, torch, dev, stream, t_q, B, topk, barrier, out_ids)
            others["mfma_tables"] = guarded(mfma_tables_workload, eng, args, torch, dev, stream, my_q, B, M, Ks, D // M, barrier)
            others["readme_n10k"] = guarded(readme_workload, args, torch, dev, arch)

            def harness_data():                                  # the main measurement's vectors and ground truth
                return keep_vectors[0], keep_vectors[1], keep_vectors[2], t_gt.cpu().numpy()
            others["ref_harness"] = guarded(ref_harness_workload, args, torch, dev, arch, barrier, harness_data)
            if args.deep_shard > 0:
                others["deep_shard"] = guarded(deep_shard_workload, args, torch, dev, arch, barrier)
            if args.deep_structured > 0:
                others["deep_structured"] = guarded(deep_structured_workload, args, torch, dev, arch, barrier)
            others["seconds_spent"] = time.perf_counter() - t_oth
            line["others"] = others
            wall["others_s"] = others["seconds_spent"]
        wall["total_s"] = time.perf_counter() - t_start
        line["wall_clock"] = wall
        emit(line, args)
    rd.close_comms()                       # (the library's own RCCL communicators: destroyed here, not at interpreter exit)
    if use_dist:
        dist.destroy_process_group()


def main_latency(args, eng, t_q, run, ivf, topk, h_tids, L, rank, world, dev):
    """Per-call latency (the reference's usage pattern: one query per call, result needed before the next call)."""
    import torch
    B = args.batch
    nq = t_q.shape[0]
    hq_all = t_q.cpu().numpy()
    n_calls = max(args.steps, 200)

    def dev_call(i):
        s = (i * B) % max(nq - B + 1, 1)
        run(t_q[s:s + B])
        torch.cuda.synchronize()

    def host_call(i):
        s = (i * B) % max(nq - B + 1, 1)
        q = hq_all[s:s + B]
        return eng.query_ivf_batch(q, topk, h_tids, L) if ivf else eng.query_linear_batch(q, topk, h_tids)

    out = {}
    for name, fn in (("device_resident", dev_call), ("host_pointer", host_call)):
        for i in range(20):
            fn(i)
        ts = []
        for i in range(n_calls):
            t0 = time.perf_counter()
            fn(i)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        out[name] = {"p50_ms": float(np.percentile(ts, 50)), "p99_ms": float(np.percentile(ts, 99)), "mean_ms": float(ts.mean())}
    if rank == 0:
        p50 = out["host_pointer"]["p50_ms"]
        print(json.dumps({"metric": "latency per call", "value": p50, "unit": "ms", "n_gpus": world, "steps": n_calls,
                          "warmup": 20, "ms_per_step": p50, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "SIFT1M-shaped %s search, N=%d, %d quer%s per call, topk=%d, fresh queries each "
                                                 "call, one synchronisation per call" % (args.workload, args.n_base, B,
                                                                                         "y" if B == 1 else "ies", topk)},
                          "latency": out,
                          "reference_readme": "0.12 ms/query at N=11k (README.md:130-140), 0.21-0.96 ms/query at N=1M on its CPU"}))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
