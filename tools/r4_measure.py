#!/usr/bin/env python3
"""Round-4 A/B in ONE process (boxes differ by a few per cent between gpurun calls): the bench shape (N = 1M, M = 32), B = 1024 and
B = 128;  fused_rerank 0 / 1: plain step, per-kernel times;  device queries -> host rows (rii_query_linear_dev_to_host) per
call;  host-pointer batch call with host_spin / host_zero_copy on and off.   usage: tools/r4_measure.py [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from rii_amd import bench_data as bd

dev = torch.device("cuda", 0)
N, M = 1_000_000, 32
base, train, query = bd.sift_like(n_base=N, n_train=100_000, n_query=4096)
cw = bd.train_pq(train, M, 256, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
eng = RiiGpu(cw, False, device=0)
eng.add_codes(codes, False)
qh = np.ascontiguousarray(query[:4096])
q = torch.from_numpy(qh).to(dev)
oi = torch.empty((4096, 1), dtype=torch.int64, device=dev)
od = torch.empty((4096, 1), dtype=torch.float32, device=dev)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
S = st.cuda_stream


def heat(fn, sec=0.2):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < sec:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


def loop(fn, K):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


out = {}
ref = {}
for rep in range(2):
    for fr in (0, 1):
        for B in (1024, 128, 256, 512):
            eng.set_option("fused_rerank", fr)
            step = lambda: eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), S)
            heat(step)
            K = 200
            ms = min(loop(step, K) for _ in range(3))
            eng.set_option("timing", 1)
            eng.timing_reset()
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            kt = {k: round(eng.timing_read(k)[0] / 50, 5) for k in ("lut", "scan", "rerank")}
            eng.set_option("timing", 0)
            ids = oi[:B].cpu().numpy().copy()
            ref.setdefault(B, ids)
            out.setdefault("dev_fr%d_B%d" % (fr, B), []).append({"ms": round(ms, 5), "kernels": kt, "ids_equal": bool((ids == ref[B]).all())})
eng.set_option("fused_rerank", 1)

# device queries -> host rows, one synchronous call per step
for B in (1024, 128):
    hi = np.empty((B, 1), np.int64)
    hd = np.empty((B, 1), np.float32)
    for spin in (1, 0):
        eng.set_option("host_spin", spin)
        call = lambda: eng.query_linear_dev_to_host(q.data_ptr(), B, 1, 0, 0, hi, hd, S)
        heat(call)
        ms = min(loop(call, 200) for _ in range(3))
        out["to_host_spin%d_B%d" % (spin, B)] = {"ms": round(ms, 5), "ids_equal": bool((hi == ref[B]).all())}
eng.set_option("host_spin", 1)

# host-pointer batch call
for B in (1024, 128):
    for spin, zc in ((0, 0), (1, 0), (1, 1)):
        eng.set_option("host_spin", spin)
        eng.set_option("host_zero_copy", 2 * zc)
        call = lambda: eng.query_linear_batch(qh[:B], 1, None)
        heat(call)
        ms = min(loop(call, 200) for _ in range(3))
        r = call()
        out["host_call_spin%d_zc%d_B%d" % (spin, zc, B)] = {"ms": round(ms, 5), "ids_equal": bool((r[0] == ref[B]).all())}
eng.set_option("host_spin", 1)
eng.set_option("host_zero_copy", 1)

# depth-2: two engines' worth of lanes is not needed -- two streams on one engine, rows to the host for both (asynchronous rows:
# pinned torch tensors as the outputs of the plain *_dev call, one synchronisation at the end of the loop)
pin_i = torch.empty((1024, 1), dtype=torch.int64).pin_memory()
pin_d = torch.empty((1024, 1), dtype=torch.float32).pin_memory()
for B in (1024, 128):
    step = lambda: eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, pin_i.data_ptr(), pin_d.data_ptr(), S)
    heat(step)
    ms = min(loop(step, 200) for _ in range(3))
    out["dev_pinned_out_async_B%d" % B] = {"ms": round(ms, 5), "ids_equal": bool((pin_i[:B].numpy() == ref[B]).all())}

js = json.dumps(out, indent=1)
print(js)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(js)
