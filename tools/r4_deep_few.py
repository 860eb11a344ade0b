#!/usr/bin/env python3
"""Few queries per call over a Deep1B-sized shard (125 M codes x M = 16 = 2 GB: a true HBM stream): which path streams fastest?
exact fp32 scan (default below fast_min_batch) vs the byte-table filter (fast_min_batch = 0) with 128 / 256 / 512 chunks.
usage: tools/r4_deep_few.py [n_codes] [out.json]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from rii_amd import bench_data as bd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
dev = torch.device("cuda", 0)
M, D = 16, 96
_, train, query = bd.sift_like(n_base=1, n_train=50_000, n_query=64, D=D, seed=99)
cw = bd.train_pq(train, M, 256, iters=5, seed=123, device=dev)
eng = RiiGpu(cw, False, device=0)
step_n = 25_000_000
rng = np.random.default_rng(1000)
for s in range(0, n, step_n):
    eng.add_codes(rng.integers(0, 256, size=(min(step_n, n - s), M), dtype=np.uint8), False)
q = torch.from_numpy(np.ascontiguousarray(query[:64])).to(dev)
oi = torch.empty((64, 1), dtype=torch.int64, device=dev); od = torch.empty((64, 1), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
out = {"n_codes": n, "bytes": n * M}
ref = {}
def run(B, topk=1):
    eng.query_linear_dev(q.data_ptr(), B, topk, 0, 0, oi.data_ptr(), od.data_ptr(), st.cuda_stream)
for name, fmb, chunks in (("exact", 33, 0), ("exact_c1024", 33, 1024), ("exact_c2048", 33, 2048), ("exact_c4096", 33, 4096), ("exact_again", 33, 0), ("filter_c0", 0, 0), ("filter_c256", 0, 256), ("filter_c512", 0, 512), ("filter_c1024", 0, 1024)):
    eng.set_option("fast_min_batch", fmb); eng.set_option("scan_chunks", chunks)
    for B in (1, 2, 4, 8, 16, 32):
        if name.startswith("exact") and B > 8: continue
        if (name.startswith("exact_c") or name == "exact_again") and B > 2: continue
        for _ in range(3): run(B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 10
        for _ in range(K): run(B)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
        eng.set_option("timing", 1); eng.timing_reset()
        for _ in range(5): run(B)
        torch.cuda.synchronize()
        scan = eng.timing_read("scan")[0] / 5
        eng.set_option("timing", 0)
        ids = oi[:B].cpu().numpy().copy()
        ref.setdefault(B, ids)
        out["%s_B%d" % (name, B)] = {"ms_per_call": round(ms, 4), "scan_ms": round(scan, 4), "stream_TBps": round(n * M / (scan * 1e-3) / 1e12, 3),
                                     "ids_equal": bool((ids == ref[B]).all())}
js = json.dumps(out, indent=1); print(js)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(js)
