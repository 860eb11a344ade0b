#!/usr/bin/env python3
"""One query per call on the README index (N=10k, D=128, M=32, Ks=256, nlist=100): host-call latency of the paths a README user hits.
Run it under `rocprofv3 --kernel-trace --stats` to see the kernels behind each number.
    tools/readme_latency.py [--calls 500] [--topk 3] [--small-topk 0|1]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rii_amd import RiiGpu
from rii_amd import bench_data as bd

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=500)
ap.add_argument("--topk", type=int, default=3)
ap.add_argument("--small-topk", type=int, default=1)
ap.add_argument("--host-spin", type=int, default=1)
ap.add_argument("--only", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
N, D, M, Ks = 10_000, 128, 32, 256
X = rng.random((N, D)).astype(np.float32)
Q = rng.random((256, D)).astype(np.float32)
cw = bd.train_pq(X[:5000], M, Ks, iters=5, seed=123, device=dev)
codes = bd.encode_pq(X, cw, device=dev)
eng = RiiGpu(cw, False, device=0)
eng.add_codes(codes, False)
eng.reconfigure(100, 5)
eng.set_option("small_topk", a.small_topk)
eng.set_option("host_spin", a.host_spin)
E = np.array([], np.int64)
sub = np.sort(rng.choice(N, 3000, replace=False)).astype(np.int64)
legs = {"linear top-1": lambda q: eng.query_linear(q, 1, E),
        "linear top-%d" % a.topk: lambda q: eng.query_linear(q, a.topk, E),
        "linear top-%d, 3000 target ids" % a.topk: lambda q: eng.query_linear(q, a.topk, sub),
        "linear top-%d, 100 target ids" % a.topk: lambda q: eng.query_linear(q, a.topk, sub[:100]),
        "ivf top-%d L=100" % a.topk: lambda q: eng.query_ivf(q, a.topk, E, 100),
        "ivf top-%d L=100, 3000 target ids" % a.topk: lambda q: eng.query_ivf(q, a.topk, sub, 100)}
for name, call in legs.items():
    if a.only and a.only not in name:
        continue
    for q in Q[:30]:
        call(q)
    ts = []
    for i in range(a.calls):
        q = Q[i % len(Q)]
        t0 = time.perf_counter()
        call(q)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    eng.set_option("timing", 1); eng.timing_reset()
    for q in Q[:50]:
        call(q)
    ks = {}
    for key in ("lut", "scan", "finalize", "ivf_fused", "ivf_exact", "bitmap", "filter_lists"):
        try:
            ms, cnt = eng.timing_read(key)
        except Exception:
            continue
        if cnt:
            ks[key] = round(ms / cnt * 1e3, 1)
    eng.set_option("timing", 0)
    print("%-40s p10 %6.1f  p50 %6.1f  p90 %6.1f us   kernels (us, HIP events): %s" % (name, np.percentile(ts, 10), np.percentile(ts, 50), np.percentile(ts, 90), ks))
