#!/usr/bin/env python3
"""Same-box A/B of an engine option on the inverted-index workloads (config 3: nlist = 1024, L = 977, B = 1024 top-1; and the README
call: N = 10k, one query per call).   usage: tools/r4_ivf_ab.py <option> <v0> <v1> ... [out.json]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from rii_amd import bench_data as bd
opt = sys.argv[1]
vals = [int(v) for v in sys.argv[2:] if v.lstrip("-").isdigit()]
outf = [v for v in sys.argv[2:] if v.endswith(".json")]
dev = torch.device("cuda", 0)
N, B, M = 1_000_000, 1024, 32
base, train, query = bd.sift_like(n_base=N, n_train=100_000, n_query=B)
cw = bd.train_pq(train, M, 256, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False); eng.reconfigure(1024, 5)
L = 977
q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev); oc = torch.empty((B,), dtype=torch.int64, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
def step(): eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
out, ref = {}, None
for rep in range(3):
    for v in vals:
        eng.set_option(opt, v)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2: step(); torch.cuda.synchronize()
        K = 300
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
        eng.set_option("timing", 2); eng.timing_reset()
        for _ in range(100): step()
        torch.cuda.synchronize()
        km = eng.timing_read("ivf_fused")[0] / 100
        eng.set_option("timing", 0)
        ids = oi.cpu().numpy().copy()
        if ref is None: ref = ids
        out.setdefault("batch_%s=%d" % (opt, v), []).append({"ms_plain": round(ms, 5), "ivf_fused_ms": round(km, 5), "ids_equal": bool((ids == ref).all())})
# the subset form of the same step (configs[3]: |S| = 100k target ids)
tids = torch.from_numpy(np.sort(np.random.default_rng(5).choice(N, 100_000, replace=False)).astype(np.int64)).to(dev)
def sstep(): eng.query_ivf_dev(q.data_ptr(), B, 1, tids.data_ptr(), tids.numel(), L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
sref = None
for rep in range(3):
    for v in vals:
        eng.set_option(opt, v)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2: sstep(); torch.cuda.synchronize()
        K = 300
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K): sstep()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
        eng.set_option("timing", 2); eng.timing_reset()
        for _ in range(100): sstep()
        torch.cuda.synchronize()
        km = eng.timing_read("ivf_fused")[0] / 100
        eng.set_option("timing", 0)
        ids = oi.cpu().numpy().copy()
        if sref is None: sref = ids
        out.setdefault("subset_%s=%d" % (opt, v), []).append({"ms_plain": round(ms, 5), "ivf_fused_ms": round(km, 5), "ids_equal": bool((ids == sref).all())})
# README call
rng = np.random.default_rng(0)
X = rng.random((10000, 128)).astype(np.float32); Q = rng.random((256, 128)).astype(np.float32)
cw2 = bd.train_pq(X[:5000], 32, 256, iters=5, seed=123, device=dev)
e2 = RiiGpu(cw2, False, device=0); e2.add_codes(bd.encode_pq(X, cw2, device=dev), False); e2.reconfigure(100, 5)
E = np.array([], np.int64)
for rep in range(3):
    for v in vals:
        e2.set_option(opt, v)
        for qq in Q[:50]: e2.query_ivf(qq, 3, E, 100)
        ts = []
        for qq in Q:
            t0 = time.perf_counter(); e2.query_ivf(qq, 3, E, 100); ts.append(time.perf_counter() - t0)
        out.setdefault("readme_%s=%d" % (opt, v), []).append(round(float(np.percentile(np.array(ts) * 1e6, 50)), 2))
js = json.dumps(out); print(js)
if outf: open(outf[0], "w").write(js)
