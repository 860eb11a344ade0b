#!/usr/bin/env python3
"""Filter path at small batches (1 - 4 query tiles) against the chunk count: step time with scan_chunks forced, SIFT-shaped N = 1M."""
import sys, time, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from rii_amd import bench_data as bd
dev = torch.device("cuda", 0)
base, train, query = bd.sift_like(n_base=1_000_000, n_train=100_000, n_query=256)
cw = bd.train_pq(train, 32, 256, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
g = RiiGpu(cw, False, device=0); g.add_codes(codes, False)
q = torch.from_numpy(np.ascontiguousarray(query)).to(dev)
oi = torch.empty((256, 1), dtype=torch.int64, device=dev); od = torch.empty((256, 1), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
out = {}
for B in (16, 32, 48, 64):
    row = {}
    for fmb, ch in ((1 << 30, 0), (0, 0), (0, 16), (0, 32), (0, 64), (0, 96), (0, 128)):
        g.set_option("fast_min_batch", fmb); g.set_option("scan_chunks", ch)
        fn = lambda: g.query_linear_dev(q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), st.cuda_stream)
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): fn()
        torch.cuda.synchronize()
        row["exact" if fmb else "filter_c%d" % ch] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
    out["B%d" % B] = row
print(json.dumps(out))
