#!/usr/bin/env python3
"""bench.py -- queries/sec + recall@1 of the batched IVFPQ query path (BASELINE.json configs[1]: SIFT1M-shaped,
D=128, M=32, Ks=256, batch=1024, top-1 linear ADC scan) on N MI355X GPUs of one node.

A step = one pass of the hot path (distance-table build + scan + top-k) over one batch of 1024 queries per GPU, inputs
and outputs resident in HBM.  N > 1: one process per GPU (torchrun).  linear / ivf / subset: the index is replicated and
every rank owns a different 1024-query batch (weak scaling); query sharding has no exchange step, so `value` carries no
collective and the line adds `with_gather` = the same loop with the all-gather of every rank's (ids, dists) rows over
RCCL/xGMI inside the timed region.  deep (configs[4] shape): the database is sharded, every rank scans its shard for the
same batch and the timed region holds the all-gather + (dist, id) merge.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel (fscan_mx_kernel / fscan_kernel for the linear scans, ivf_fused_kernel for the inverted index) timed by
                HIP events recorded around each of its launches in the timed region, against the resource that binds it:
                the LDS table-gather rate for the scans (157.3 TB/s = ds_read_b128 256 B/clk/CU x 256 CUs x 2.4 GHz), HBM
                for the inverted index and the single-query Deep scan.  `hbm` inside it holds the counter traffic.
  host_call     the same step through the host-pointer C ABI (H2D of the queries + D2H of ids/dists included: SURVEY 8d)
  cpu_baseline  the real reference build (oracle/_ref; else the C oracle) on this box's host cores, same index and
                queries, with `ids_match_gpu`; rank 0, N=1 only.
`--latency`: one call per step with fresh queries and a synchronisation per call (the reference's usage pattern).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LDS_PEAK_GBPS = 157286.4         # ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X guide, LDS table)
HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n-base", type=int, default=1_000_000)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--workload", default="linear", choices=["linear", "ivf", "subset", "subset-ivf", "deep"],
                    help="linear/ivf/subset/subset-ivf: BASELINE configs[1..3] (SIFT1M-shaped, index replicated, queries "
                         "sharded); deep: configs[4] shape (D=96, M=16, --n-base codes PER GPU, database sharded)")
    ap.add_argument("--topk", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-call", action="store_true")
    ap.add_argument("--no-fresh", action="store_true",
                    help="skip the extra measurement with a different query batch every step")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra two-stream measurement (reported beside `value`, never as it)")
    ap.add_argument("--latency", action="store_true",
                    help="per-call latency: fresh queries every call, one synchronisation per call (use with --batch 1)")
    ap.add_argument("--lut-mode", default="exact", choices=["exact", "mfma"])
    ap.add_argument("--scan-order", type=int, default=1, choices=[0, 1],
                    help="1: scan the LDS-friendly permutation of the codes (default), 0: id order")
    ap.add_argument("--scan-mx", type=int, default=1, choices=[0, 1],
                    help="1 (default): the filter scan of the M = 16 / 32 shapes sums its table bytes on the matrix cores "
                         "(fscan_mx_kernel); 0: on the vector ALU (fscan_kernel)")
    ap.add_argument("--scan-mode", type=int, default=1, choices=[0, 1],
                    help="1: 8-bit filter + exact re-rank (default), 0: exact scan of every code; identical results")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference itself (oracle/_ref) on the host cores, same index, same queries
# --------------------------------------------------------------------------------------------------------------------
def cpu_baseline(workload, eng, cw, codes, queries, topk, tids, L, arch_hint):
    """The reference's own path, one query per call (it has no batch entry point): QueryLinear is OpenMP-parallel over
    N (src/rii.h:213,222), QueryIvf single-threaded (rii.h:244-326).  OpenMP's default thread count (all hardware
    threads) is not the reference's best setting on a many-core host, so a few counts are timed on a bounded sample and
    the BEST is reported with `cores` = the threads it used.  For the inverted-index workloads the reference receives the
    GPU engine's centres and posting lists through its own py::pickle set-state (src/main.cpp:39-52)."""
    import ctypes
    from oracle import oracle as O
    ref, arch, flav = O.load_reference()
    E = np.array([], np.int64)
    tids = E if tids is None else tids
    ivf = workload in ("ivf", "subset-ivf")
    if ref is not None:
        kind = "reference"
        if ivf:
            ref_e = ref.RiiCpp.__new__(ref.RiiCpp)
            ref_e.__setstate__(eng.__getstate__())
        else:
            ref_e = ref.RiiCpp(cw, False)
            ref_e.add_codes(codes, False)
    else:
        kind = "port"
        ref_e = O.OracleRii(cw, False, simd_arch=arch_hint)
        ref_e.add_codes(codes, False)
        if ivf:
            ref_e.centers = np.array(eng.coarse_centers, np.uint8)
            ref_e._lists = eng.posting_lists

    def call(q):
        return ref_e.query_ivf(q, topk, tids, L) if ivf else ref_e.query_linear(q, topk, tids)

    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    ncpu = os.cpu_count() or 1
    settings = [1] if ivf else (sorted({ncpu, min(ncpu, 64), min(ncpu, 16), min(ncpu, 8)}, reverse=True) if gomp else [ncpu])
    tried, best, ids_full = [], None, None
    for nthr in settings:
        if gomp:
            gomp.omp_set_num_threads(int(nthr))
        call(queries[0])                                              # warm-up
        t0 = time.perf_counter()
        for q in queries[:4]:
            call(q)
        per = (time.perf_counter() - t0) / 4
        n = int(min(len(queries), max(8, 6.0 / max(per, 1e-6))))      # ~6 s of CPU work per setting
        t0 = time.perf_counter()
        res = [call(q)[0] for q in queries[:n]]
        dt = time.perf_counter() - t0
        tried.append("%d threads: %.1f q/s over %d queries" % (nthr, n / dt, n))
        if ids_full is None or len(res) > len(ids_full):
            ids_full = res
        if best is None or n / dt > best[0]:
            best = (n / dt, nthr, n)
    what = {"linear": "full %d-code linear scan" % codes.shape[0], "subset": "linear scan of %d target ids" % len(tids),
            "ivf": "inverted index nlist=1024 L=%d" % L, "subset-ivf": "inverted index nlist=1024 L=%d over %d target ids" % (L, len(tids))}
    return {"value": best[0], "unit": "queries/s", "ms_per_query": 1e3 / best[0], "cores": best[1], "kind": kind,
            "sample": "one query per call, %s, top-%d%s; thread counts tried: %s"
                      % (what[workload], topk, (", build flavour " + flav) if ref is not None else "", "; ".join(tried))}, ids_full


# --------------------------------------------------------------------------------------------------------------------
# roofline bookkeeping
# --------------------------------------------------------------------------------------------------------------------
def profile_table(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def filter_kernel_name(args, M):
    return "fscan_mx_kernel" if (args.scan_mx and M in (16, 32, 64)) else "fscan_kernel"


def workload_key(args, n_scanned):
    key = "%s/scan_mode=%d/M=%d/N=%d/B=%d" % (args.workload, args.scan_mode, args.M if args.workload != "deep" else 16,
                                              n_scanned, args.batch)
    if args.scan_mode == 1 and not args.scan_mx:
        key += "/scan_mx=0"
    if args.topk != 1:
        key += "/topk=%d" % args.topk
    return key


def roofline_scan(args, kernel_name, B, n_codes, M, Ks, avg_s, launches, steps, byte_tables, pmc_key):
    """Linear scans: the binding resource is the LDS table gather -- every (query, code, m) lookup reads one table entry
    (1 byte from the filter's byte tables, 4 bytes from the exact fp32 tables) out of LDS; the code bytes themselves are
    shared by the whole batch through L2/LDS, so HBM sees ~N*M bytes per launch, not B*N*M."""
    lookups = B * n_codes * M                                  # M table gathers per (query, code)
    entry = 1 if byte_tables else 4
    achieved = lookups * entry / avg_s / 1e9 if avg_s > 0 else 0.0
    pmc = profile_table("pmc.json").get(pmc_key, {})
    traffic = profile_table("traffic.json").get(pmc_key, {}).get("hbm_bytes_per_launch")
    # codes once (fscan_kernel's formatted copy of the M = 16 / 32 shapes holds 2 bytes per code byte; fscan_mx_kernel's is a
    # permutation of the code bytes) + the tables staged
    fmt = 2 if (byte_tables and M in (16, 32) and Ks == 256 and not args.scan_mx) else 1      # (M = 64: plain or permuted code bytes)
    floor = n_codes * M * fmt + B * M * Ks * entry
    hbm = {"algorithmic_bytes_per_launch": lookups, "compulsory_floor_bytes": floor, "traffic_bytes": traffic,
           "achieved": (traffic / avg_s / 1e9) if (traffic and avg_s > 0) else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    if hbm["achieved"] is not None:
        hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBPS
    return {"bound": "lds-gather", "kernel": kernel_name, "achieved": achieved, "peak": LDS_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / LDS_PEAK_GBPS, "traffic": traffic, "avg_launch_ms": avg_s * 1e3, "launches": launches,
            "launches_per_step": launches / max(steps, 1), "table_lookups_per_launch": lookups, "entry_bytes": entry,
            "conflict_frac": pmc.get("lds_conflict_frac"), "lds_busy": pmc.get("lds_busy"), "valu_busy": pmc.get("valu_busy"),
            "mfma_busy": pmc.get("mfma_busy"), "shader_clock_ghz": pmc.get("shader_clock_ghz"),
            "hbm": hbm,
            "note": "achieved = table-entry bytes gathered from LDS per second (B*N*M lookups x entry_bytes / kernel time); "
                    "peak = conflict-free ds_read_b128 rate at the nominal 2.4 GHz; conflict_frac = SQ_LDS_BANK_CONFLICT / "
                    "SQ_LDS_IDX_ACTIVE, lds_busy / valu_busy / mfma_busy = busy cycles over the launch's shader cycles, from the "
                    "rocprofv3 --pmc passes of the same command (profiles/)"}


def roofline_hbm(kernel_name, alg_bytes, avg_s, launches, steps, traffic):
    achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
    return {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": avg_s * 1e3, "launches": launches, "launches_per_step": launches / max(steps, 1)}


def timed_loop(fn, steps, barrier):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    return time.perf_counter() - t0


# --------------------------------------------------------------------------------------------------------------------
def main_deep(args, world, rank, local, dev, arch):
    """Deep1B-shaped database sharding (BASELINE configs[4]): D=96, M=16, Ks=256; every rank holds --n-base codes
    (uniform random bytes: throughput only, so no recall), all ranks answer the SAME batch on their shard, global id =
    shard offset + local id, results all-gathered over RCCL and merged on the device under the (dist, id) rule."""
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
    out_ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
    out_d = torch.empty((B, topk), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(device=dev)      # engine kernels, RCCL calls and the merge are ordered on ONE torch stream
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    offset = rank * n_shard
    merged = [None]
    use_dist = dist.is_initialized()

    def step():
        eng.query_linear_dev(q.data_ptr(), B, topk, 0, 0, out_ids.data_ptr(), out_d.data_ptr(), stream)
        merged[0] = rd.allgather_merge_topk(out_ids, out_d, topk, id_offset=offset)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # inside the timed region only the dominant kernel carries HIP events (2 records per step); the other kernels' shares
    # come from a short untimed pass afterwards (events around every launch cost ~10 % of a 0.4 ms step)
    eng.set_option("timing", 2)
    eng.timing_reset()
    elapsed = timed_loop(step, args.steps, barrier)
    eng.set_option("timing", 0)
    dom = {kn: eng.timing_read(kn) for kn in ("scan", "ivf_fused", "ivf_scan")}
    eng.timing_reset()
    eng.set_option("timing", 1)
    n_break = max(3, min(args.steps, 10))
    for _ in range(n_break):
        step()
    torch.cuda.synchronize()
    eng.set_option("timing", 0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms, k_n = dom["scan"]
    if rank == 0:
        avg_s = (k_ms / max(args.steps, 1)) * 1e-3
        filt = bool(args.scan_mode and (topk > 1 or B >= eng.get_option("fast_min_batch")))
        key = workload_key(args, n_shard)
        if B == 1:
            roof = roofline_hbm("scan_kernel", n_shard * M, avg_s, k_n, args.steps,
                                profile_table("traffic.json").get(key, {}).get("hbm_bytes_per_launch"))
        else:
            roof = roofline_scan(args, filter_kernel_name(args, M) if filt else "scan_kernel", B, n_shard, M, Ks, avg_s, k_n, args.steps,
                                 filt, key)
        print(json.dumps({
            "metric": "queries/sec", "value": B * args.steps / elapsed, "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Deep1B-shaped linear ADC scan, D=96 M=16 Ks=256, %d codes per GPU (database sharded, "
                                   "%d codes total), batch=%d, topk=%d" % (n_shard, n_shard * world, B, topk),
                       "global_batch": B, "parallelism": "database-sharded x%d, RCCL all-gather + device (dist,id) merge in the timed region" % world,
                       "scan_mode": "byte-table filter + exact fp32 re-rank" if args.scan_mode else "exact fp32 scan"},
            "recall_at_1": None, "roofline": roof}))
    if use_dist:
        dist.destroy_process_group()


def main():
    args = parse()
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
    assert world == args.gpus, "launch with torchrun --nproc-per-node %d (got WORLD_SIZE=%d)" % (args.gpus, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, M, Ks, D = args.batch, args.M, 256, 128
    arch = host_simd_arch()
    if args.workload == "deep":
        return main_deep(args, world, rank, local, dev, arch)

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
    if ivf:
        eng.reconfigure(1024, 5)
        L = int(np.round(N / 1024))
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

    def run(qt):
        if ivf:
            eng.query_ivf_dev(qt.data_ptr(), qt.shape[0], topk, d_tids, S, L, out_ids.data_ptr(), out_d.data_ptr(),
                              out_cnt.data_ptr(), stream)
        else:
            eng.query_linear_dev(qt.data_ptr(), qt.shape[0], topk, d_tids, S, out_ids.data_ptr(), out_d.data_ptr(), stream)

    def step():
        run(my_q)       # query sharding: every rank owns the results of its own queries, no exchange

    gathered = [None]

    def step_gather():
        run(my_q)
        gathered[0] = rd.allgather_query_shards(out_ids, out_d)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.latency:
        return main_latency(args, eng, t_q, run, ivf, topk, h_tids, L, rank, world, dev)

    for _ in range(args.warmup):
        step()
    barrier()
    # inside the timed region only the dominant kernel carries HIP events (2 records per step); the other kernels' shares
    # come from a short untimed pass afterwards (events around every launch cost ~10 % of a 0.4 ms step)
    eng.set_option("timing", 2)
    eng.timing_reset()
    elapsed = timed_loop(step, args.steps, barrier)
    eng.set_option("timing", 0)
    dom = {kn: eng.timing_read(kn) for kn in ("scan", "ivf_fused", "ivf_scan")}
    eng.timing_reset()
    eng.set_option("timing", 1)
    n_break = max(3, min(args.steps, 10))
    for _ in range(n_break):
        step()
    torch.cuda.synchronize()
    eng.set_option("timing", 0)
    res_ids = out_ids.cpu().numpy().copy()
    res_cnt = out_cnt.cpu().numpy().copy() if ivf else None
    elapsed_g = None
    if use_dist:
        for _ in range(args.warmup):
            step_gather()
        elapsed_g = timed_loop(step_gather, args.steps, barrier)
        cdev = "cpu" if host_coll else dev
        t = torch.tensor([elapsed, elapsed_g], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_g = float(t[0].item()), float(t[1].item())
        allq = gathered[0][0]                        # the gathered batch really is every rank's rows, in rank order
        assert allq.shape[0] == B * world and torch.equal(allq[rank * B:(rank + 1) * B].to(out_ids.device), out_ids)

    kernel = "scan"
    if ivf:
        kernel = "ivf_fused" if eng.get_option("ivf_fused") else "ivf_scan"
    k_ms, k_n = dom[kernel]
    extra = {}
    for kn in ("lut", "quant", "rerank", "kth", "tie", "select", "gather", "ivf_exact", "ivf_coarse", "ivf_plan", "ivf_scan", "ivf_select"):
        if kn == kernel:
            continue
        ms_, n_ = eng.timing_read(kn)
        if n_:
            extra[kn + "_ms_per_step"] = ms_ / n_break
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
        key = workload_key(args, n_scanned)
        if ivf:
            # coarse codes + visited posting ids + L gathered codes per query
            w = min(1024, int(np.round(L * 1024 / (S if S else N))) + 3)
            alg = B * (1024 * M + w * (N // 1024) * 4 + L * M)
            roof = roofline_hbm("ivf_fused_kernel", alg, avg_s, k_n, args.steps,
                                profile_table("traffic.json").get(key, {}).get("hbm_bytes_per_launch"))
            roof["note"] = ("latency-bound random 32-byte gathers: algorithmic bytes = B*(nlist*M + w*mean_list_len*4 + L*M)")
        else:
            filt = bool(args.scan_mode and (topk > 1 or B >= eng.get_option("fast_min_batch")))
            roof = roofline_scan(args, filter_kernel_name(args, M) if filt else "scan_kernel", B, n_scanned, M, Ks, avg_s, k_n, args.steps,
                                 filt, key)
        roof.update(extra)
        line = {
            "metric": "queries/sec", "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SIFT1M-shaped %s search, D=128 M=%d Ks=256, N=%d, batch=%d per GPU, topk=%d%s%s"
                                   % (args.workload, M, N, B, topk, (", nlist=1024 L=%d" % L) if L else "",
                                      (", |target_ids|=%d" % S) if S else ""),
                       "global_batch": B * world,
                       "parallelism": "query-sharded x%d, index replicated; value: no exchange step (results stay with the "
                                      "owning rank), with_gather: RCCL all-gather of the result rows in the timed region" % world,
                       "lut_mode": args.lut_mode, "simd_order": arch,
                       "scan_mode": "8-bit filter + exact fp32 re-rank" if args.scan_mode else "exact fp32 scan"},
            "recall_at_1": recall,
            "roofline": roof,
        }
        if elapsed_g is not None:
            line["with_gather"] = {"ms_per_step": elapsed_g / args.steps * 1e3, "value": B * world * args.steps / elapsed_g,
                                   "unit": "queries/s", "backend": dist.get_backend(),
                                   "collective": "all_gather of %d B per rank (ids int64 + dists f32), device tensors" % (B * topk * 12),
                                   "collective_share": max(0.0, 1.0 - elapsed / elapsed_g)}
        if host is not None:
            line["host_call"] = host
        if fresh is not None:
            line["fresh_queries"] = fresh
        if pipe is not None:
            line["pipelined"] = pipe
        if world == 1 and not args.no_cpu_baseline:
            cb, cpu_res = cpu_baseline(args.workload, eng, cw, codes, my_q.cpu().numpy(), topk, h_tids, L, arch)
            n = len(cpu_res)
            if ivf:
                ok = all(list(res_ids[b, :int(res_cnt[b])]) == list(cpu_res[b]) for b in range(n))
            else:
                ok = all(list(res_ids[b]) == list(cpu_res[b]) for b in range(n))
            cb["ids_match_gpu"] = bool(ok)
            cb["queries_compared"] = n
            line["cpu_baseline"] = cb
        print(json.dumps(line))
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
