#!/usr/bin/env python3
"""Within-one-process A/B of an integer engine option on the bench shape (boxes differ by a few per cent between gpurun calls):
   tools/opt_ab.py <option> <value> <value> ...   -> plain step ms, scan-kernel ms at B = 1024 and B = 128, interleaved twice."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from rii_amd import bench_data as bd
opt, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]        # RII_AB_BATCHES=1024,128: batch sizes
dev = torch.device("cuda", 0)
N, B, M = 1_000_000, 4096, 32
base, train, query = bd.sift_like(n_base=N, n_train=100_000, n_query=B)
cw = bd.train_pq(train, M, 256, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
def step(b=B): eng.query_linear_dev(q.data_ptr(), b, 1, 0, 0, oi.data_ptr(), od.data_ptr(), st.cuda_stream)
out, ref = {}, {}
for rep in range(3):
    for v in vals:
        for b in [int(x) for x in os.environ.get("RII_AB_BATCHES", "1024,128").split(",")]:
            eng.set_option(opt, v)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.25: step(b); torch.cuda.synchronize()
            eng.set_option("timing", 2); eng.timing_reset()
            K = 60
            for _ in range(K): step(b)
            torch.cuda.synchronize()
            scan = eng.timing_read("scan")[0] / K
            eng.set_option("timing", 0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K): step(b)
            torch.cuda.synchronize(); el0 = (time.perf_counter() - t0) / K
            ids = oi[:b].cpu().numpy().copy()
            ref.setdefault(b, ids)
            out.setdefault("%s=%d_B%d" % (opt, v, b), []).append({"ms_plain": round(el0 * 1e3, 5), "scan_ms": round(scan, 5), "ids_equal": bool((ids == ref[b]).all())})
print(json.dumps(out))
