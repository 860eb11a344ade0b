#!/usr/bin/env python3
"""fscan_mx_kernel time against N for a small batch (B = 128 and 16: 8 / 1 query tiles): the fixed part of a launch (table staging, first
thresholds, flush) vs the part that scales with the codes.   usage: tools/r4_scan_vs_n.py [out.json]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from rii_amd import bench_data as bd
dev = torch.device("cuda", 0)
M = 32
base, train, query = bd.sift_like(n_base=4_000_000, n_train=100_000, n_query=1024)
cw = bd.train_pq(train, M, 256, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
q = torch.from_numpy(np.ascontiguousarray(query)).to(dev)
oi = torch.empty((1024, 1), dtype=torch.int64, device=dev); od = torch.empty((1024, 1), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
out = {}
for N in (125_000, 250_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes[:N], False)
    for B in (16, 128, 1024):
        def step(): eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), st.cuda_stream)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15: step(); torch.cuda.synchronize()
        K = 100
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
        eng.set_option("timing", 1); eng.timing_reset()
        for _ in range(50): step()
        torch.cuda.synchronize()
        kt = {k: round(eng.timing_read(k)[0] / 50 * 1e3, 2) for k in ("lut", "scan", "rerank")}
        eng.set_option("timing", 0)
        out["N%d_B%d" % (N, B)] = {"step_us": round(ms * 1e3, 2), "kernels_us": kt}
    del eng
js = json.dumps(out, indent=0); print(js)
if len(sys.argv) > 1: open(sys.argv[1], "w").write(js)
