#!/usr/bin/env python3
"""bench.py -- queries/sec + recall@1 of the batched IVFPQ linear ADC scan (BASELINE.json configs[1]:
SIFT1M-shaped, D=128, M=32, Ks=256, batch=1024, top-1) on N MI355X GPUs of one node.

A step = one pass of the hot path (distance-table build + linear ADC scan + top-1) over one batch of 1024
queries per GPU, inputs and outputs resident in HBM.  N > 1: one process per GPU (torchrun), the index is
replicated, every rank owns a different 1024-query batch (weak scaling) and the per-rank results are
all-gathered over RCCL/xGMI inside the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line: `roofline` (the dominant kernel, scan_kernel, from HIP events recorded around each
of its launches during the timed region) and `cpu_baseline` (the real reference build oracle/_ref, or the C
oracle, timed on this box's host cores on a bounded sample of the same workload; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--n-base", type=int, default=1_000_000)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--workload", default="linear", choices=["linear", "ivf", "subset", "deep"],
                    help="linear/ivf/subset: BASELINE configs[1..3] (SIFT1M-shaped, index replicated, queries sharded); "
                         "deep: configs[4] shape (D=96, M=16, --n-base codes PER GPU, database sharded, all-gather + top-k merge)")
    ap.add_argument("--topk", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lut-mode", default="exact", choices=["exact", "mfma"])
    ap.add_argument("--scan-order", type=int, default=1, choices=[0, 1],
                    help="1: scan the LDS-friendly permutation of the codes (default), 0: id order")
    ap.add_argument("--scan-mode", type=int, default=1, choices=[0, 1],
                    help="1: 8-bit filter + exact re-rank (default), 0: exact scan of every code; identical results")
    return ap.parse_args()


def cpu_baseline(cw, codes, queries, arch_hint):
    """The reference's own path (per-query loop, OpenMP over N: src/rii.h:195-242) on the host cores.  The reference
    runs with OpenMP's default thread count (= all hardware threads); on a many-core host that is not its best setting
    (the serial std::partial_sort + 16 MB resize dominate and fork/join over 256 threads costs), so a few thread counts
    are timed on a bounded sample and the BEST one is reported, with `cores` = the threads it used."""
    import ctypes
    from oracle import oracle as O
    ref, arch, flav = O.load_reference()
    E = np.array([], np.int64)
    if ref is not None:
        eng = ref.RiiCpp(cw, False)
        kind = "reference"
    else:
        eng = O.OracleRii(cw, False, simd_arch=arch_hint)
        kind = "port"
    eng.add_codes(codes, False)
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    ncpu = os.cpu_count() or 1
    settings = sorted({ncpu, min(ncpu, 64), min(ncpu, 16), min(ncpu, 8)}, reverse=True) if gomp else [ncpu]
    tried, best = [], None
    ids_full = None
    for nthr in settings:
        if gomp:
            gomp.omp_set_num_threads(int(nthr))
        eng.query_linear(queries[0], 1, E)                        # warm-up
        t0 = time.perf_counter()
        for q in queries[:4]:
            eng.query_linear(q, 1, E)
        per = (time.perf_counter() - t0) / 4
        n = int(min(len(queries), max(8, 6.0 / max(per, 1e-6))))  # ~6 s of CPU work per setting
        t0 = time.perf_counter()
        ids = [eng.query_linear(q, 1, E)[0][0] for q in queries[:n]]
        dt = time.perf_counter() - t0
        tried.append("%d threads: %.1f q/s over %d queries" % (nthr, n / dt, n))
        if ids_full is None or len(ids) > len(ids_full):
            ids_full = ids
        if best is None or n / dt > best[0]:
            best = (n / dt, nthr, n)
    return {"value": best[0], "unit": "queries/s", "cores": best[1], "kind": kind,
            "sample": "one query per call (the reference has no batch entry point), full %d-code linear scan, top-1%s; "
                      "thread counts tried: %s" % (codes.shape[0], (", build flavour " + flav) if ref is not None else "",
                                                    "; ".join(tried))}, np.array(ids_full)


def lds_gather(args, alg_bytes, avg_s, kernel):
    """The roofline that actually binds the linear scan: table-entry bytes gathered from LDS per second against the
    ds_read_b128 peak (256 B/clk/CU x 256 CUs x 2.4 GHz = 157.3 TB/s; MI355X guide, LDS table).  One (query, code, m)
    lookup moves 1 byte with the byte-table filter and 4 bytes with the exact fp32 scan; random code bytes cost ~2.8x the
    conflict-free cycles (SQ_LDS_BANK_CONFLICT), so ~0.36 is the practical ceiling of this formulation."""
    if kernel != "scan" or avg_s <= 0:
        return None
    entry_bytes = 1 if (args.scan_mode and (args.topk > 1 or args.batch >= 128)) else 4
    achieved = alg_bytes * entry_bytes / avg_s / 1e12
    return {"achieved": achieved, "peak": 157.3, "unit": "TB/s", "frac": achieved / 157.3, "entry_bytes": entry_bytes}


def measured_traffic_key(key):
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def measured_traffic(args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/*_traffic.json:
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per the MI355X guide's gfx950 correction), or None if not measured for this
    exact workload/mode."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        tab = json.load(open(path))
    except Exception:
        return None
    key = "%s/scan_mode=%d/M=%d/N=%d/B=%d" % (args.workload, args.scan_mode, args.M, args.n_base, args.batch)
    if args.topk != 1:
        key += "/topk=%d" % args.topk
    return tab.get(key, {}).get("hbm_bytes_per_launch")


def main_deep(args, world, rank, local, dev, arch):
    """Deep1B-shaped database sharding (BASELINE configs[4]): D=96, M=16, Ks=256; every rank holds --n-base codes
    (uniform random bytes: throughput only, so no recall), all ranks answer the SAME batch on their shard, global id =
    shard offset + local id, results all-gathered over RCCL and merged under the canonical (dist, id) rule."""
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
    del codes
    topk = args.topk
    q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
    out_ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
    out_d = torch.empty((B, topk), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    offset = rank * n_shard
    merged = [None]

    def step():
        eng.query_linear_dev(q.data_ptr(), B, topk, 0, 0, out_ids.data_ptr(), out_d.data_ptr(), stream)
        merged[0] = rd.allgather_merge_topk(out_ids, out_d, topk, id_offset=offset)

    use_dist = dist.is_initialized()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.set_option("timing", 1)
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms, k_n = eng.timing_read("scan")
    if rank == 0:
        alg_bytes = B * n_shard * M
        avg_s = (k_ms / max(k_n, 1)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        print(json.dumps({
            "metric": "queries/sec", "value": B * args.steps / elapsed, "unit": "queries/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Deep1B-shaped linear ADC scan, D=96 M=16 Ks=256, %d codes per GPU (database sharded, "
                                   "%d codes total), batch=%d, topk=%d" % (n_shard, n_shard * world, B, topk),
                       "global_batch": B, "parallelism": "database-sharded x%d, all-gather + (dist,id) merge" % world,
                       "scan_mode": "byte-table filter + exact fp32 re-rank" if args.scan_mode else "exact fp32 scan"},
            "recall_at_1": None,
            "roofline": {"bound": "hbm", "kernel": "fscan_kernel" if (args.scan_mode and (topk > 1 or B >= 128)) else "scan_kernel",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": measured_traffic_key("deep/scan_mode=%d/M=16/N=%d/B=%d" % (args.scan_mode, n_shard, B)),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_s * 1e3, "launches": k_n}}))
    if use_dist:
        dist.destroy_process_group()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from rii_amd import RiiGpu, host_simd_arch
    from rii_amd import bench_data as bd

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
    if rank == 0:
        base, train, query = bd.sift_like(n_base=N, n_train=100_000, n_query=max(10_000, B * world), D=D)
        cw = bd.train_pq(train, M, Ks, iters=10, seed=123, device=dev)
        codes = bd.encode_pq(base, cw, device=dev)
        gt = bd.exact_nn(base, query[:B * world], device=dev)
        t_cw = torch.from_numpy(cw).to(dev)
        t_codes = torch.from_numpy(codes).to(dev)
        t_q = torch.from_numpy(np.ascontiguousarray(query[:B * world])).to(dev)
        t_gt = torch.from_numpy(gt).to(dev)
        del base, train
    else:
        t_cw = torch.empty((M, Ks, D // M), dtype=torch.float32, device=dev)
        t_codes = torch.empty((N, M), dtype=torch.uint8, device=dev)
        t_q = torch.empty((B * world, D), dtype=torch.float32, device=dev)
        t_gt = torch.empty((B * world,), dtype=torch.int64, device=dev)
    host_coll = use_dist and dist.get_backend() != "nccl"         # gloo: run the collectives on host copies

    def bcast(t):
        if host_coll:
            c = t.cpu()
            dist.broadcast(c, 0)
            t.copy_(c)
        else:
            dist.broadcast(t, 0)

    def all_gather(outs, t):
        if host_coll:
            hs = [torch.empty_like(t, device="cpu") for _ in outs]
            dist.all_gather(hs, t.cpu())
            for o, h in zip(outs, hs):
                o.copy_(h)
        else:
            dist.all_gather(outs, t)

    if use_dist:
        for t in (t_cw, t_codes, t_q, t_gt):
            bcast(t)
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
    topk = args.topk
    S, L = 0, 0
    d_tids = 0
    if args.workload != "linear":
        eng.reconfigure(1024, 5)
        L = int(np.round(N / 1024))
    if args.workload == "subset":
        rng = np.random.default_rng(7)
        tids = torch.from_numpy(np.sort(rng.choice(N, 100_000, replace=False)).astype(np.int64)).to(dev)
        S, d_tids = tids.numel(), tids.data_ptr()
    out_ids = torch.empty((B, topk), dtype=torch.int64, device=dev)
    out_d = torch.empty((B, topk), dtype=torch.float32, device=dev)
    out_cnt = torch.empty((B,), dtype=torch.int64, device=dev)
    gather_ids = [torch.empty_like(out_ids) for _ in range(world)] if use_dist else None
    gather_d = [torch.empty_like(out_d) for _ in range(world)] if use_dist else None
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        if args.workload == "ivf":
            eng.query_ivf_dev(my_q.data_ptr(), B, topk, d_tids, S, L, out_ids.data_ptr(), out_d.data_ptr(),
                              out_cnt.data_ptr(), stream)
        else:
            eng.query_linear_dev(my_q.data_ptr(), B, topk, d_tids, S, out_ids.data_ptr(), out_d.data_ptr(), stream)
        # query sharding has no exchange step: every rank owns the results of its own queries, so the
        # timed region carries no collective -- only the barrier on both sides

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.set_option("timing", 1)
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    eng.set_option("timing", 0)
    if use_dist:           # untimed: the optional gather of all ranks' rows (12 KB per rank at top-1) still has to work
        all_gather(gather_ids, out_ids)
        all_gather(gather_d, out_d)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_coll else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel = "scan"
    if args.workload == "ivf":
        kernel = "ivf_fused" if eng.get_option("ivf_fused") else "ivf_scan"
    k_ms, k_n = eng.timing_read(kernel)
    lut_ms, lut_n = eng.timing_read("lut")
    extra = {}
    for kn in ("quant", "rerank", "kth", "select", "gather", "ivf_coarse", "ivf_plan", "ivf_scan", "ivf_select"):
        ms_, n_ = eng.timing_read(kn)
        if n_:
            extra[kn + "_avg_launch_ms"] = ms_ / n_
    recall = bd.recall_at_r(out_ids.cpu().numpy(), my_gt, 1)
    if use_dist:
        r = torch.tensor([recall], dtype=torch.float64, device="cpu" if host_coll else dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        recall = float(r.item()) / world
        allq = torch.cat(gather_ids, dim=0)          # the gathered batch really is every rank's rows, in rank order
        assert allq.shape[0] == B * world and torch.equal(allq[rank * B:(rank + 1) * B], out_ids)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        qps = B * world * args.steps / elapsed
        n_scanned = S if args.workload == "subset" else N
        if args.workload == "ivf":
            alg_bytes = B * (1024 * M + 4 * L * 4 + L * M)        # coarse codes + ids + L codes
        else:
            alg_bytes = B * n_scanned * M                          # M code bytes per (query, code)
        # dominant-kernel time per step: top-k runs the scan kernel twice per step (a sampled pass 1 + pass 2), and the
        # algorithmic bytes of the step are charged against both together
        avg_s = (k_ms / max(args.steps, 1)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        line = {
            "metric": "queries/sec", "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SIFT1M-shaped %s ADC scan, D=128 M=%d Ks=256, N=%d, batch=%d per GPU, topk=%d%s"
                                   % (args.workload, M, N, B, topk, (", nlist=1024 L=%d" % L) if L else ""),
                       "global_batch": B * world, "parallelism": "query-sharded x%d, index replicated, no collective in the timed region" % world,
                       "lut_mode": args.lut_mode, "simd_order": arch,
                       "scan_mode": "8-bit filter + exact fp32 re-rank" if args.scan_mode else "exact fp32 scan"},
            "recall_at_1": recall,
            "roofline": {"bound": "hbm", "kernel": ("fscan" if (args.scan_mode and kernel == "scan") else kernel) + "_kernel", "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": measured_traffic(args),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_s * 1e3, "launches": k_n,
                         "launches_per_step": k_n / max(args.steps, 1),
                         "lut_avg_launch_ms": lut_ms / max(lut_n, 1), **extra,
                         "lds_gather": lds_gather(args, alg_bytes, avg_s, kernel),
                         "note": "codes are shared by the whole batch through LDS/L2, so algorithmic bytes exceed HBM "
                                 "traffic by design; the scan's limiters are the LDS gather rate (bank conflicts) and VALU issue, "
                                 "both ~70-80% busy (profiles/r01_fscan_pmc_counters.txt: SQ_LDS_IDX_ACTIVE, "
                                 "SQ_LDS_BANK_CONFLICT, SQ_INSTS_VALU)"},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "linear" and topk == 1:
            cb, cpu_ids = cpu_baseline(cw, codes, my_q.cpu().numpy(), arch)
            gpu_ids = out_ids.cpu().numpy()[:len(cpu_ids), 0]
            cb["ids_match_gpu"] = bool(np.array_equal(cpu_ids, gpu_ids))
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
