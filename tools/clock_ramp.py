#!/usr/bin/env python3
"""Does the step time depend on how long the GPU has been busy (clock ramp), on the timing events, or on re-submitting the
same batch?  Prints ms/step of consecutive 20-step loops: plain, with timing=2 events, fresh batches, after an idle gap."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rii_amd import RiiGpu, host_simd_arch
from rii_amd import bench_data as bd

N, B, M = 1_000_000, 1024, 32
dev = torch.device("cuda", 0)
base, train, query = bd.sift_like(n_base=N, n_train=100_000, n_query=8 * B)
cw = bd.train_pq(train, M, 256, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
eng = RiiGpu(cw, False, simd_arch=host_simd_arch(), device=0)
eng.add_codes(codes, False)
qs = [torch.from_numpy(np.ascontiguousarray(query[i * B:(i + 1) * B])).to(dev) for i in range(8)]
oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev); torch.cuda.set_stream(side); st = side.cuda_stream

def loop(K, fresh=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K):
        q = qs[i % 8] if fresh else qs[0]
        eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3

out = {}
loop(3)
time.sleep(1.0)
out["cold_then_consecutive_plain_20"] = [round(loop(20), 4) for _ in range(12)]
eng.set_option("timing", 2)
out["consecutive_timing2_20"] = [round(loop(20), 4) for _ in range(6)]
eng.set_option("timing", 0); eng.timing_reset()
out["consecutive_fresh_20"] = [round(loop(20, True), 4) for _ in range(6)]
out["plain_again_20"] = [round(loop(20), 4) for _ in range(4)]
time.sleep(1.0)
out["after_1s_idle_plain_5_each"] = [round(loop(5), 4) for _ in range(12)]
out["long_plain_400"] = round(loop(400), 4)
out["long_fresh_400"] = round(loop(400, True), 4)
eng.set_option("timing", 2)
out["long_timing2_400"] = round(loop(400), 4)
print(json.dumps(out))
