#!/usr/bin/env python3
"""Where does ivf_fused_kernel's time go?  Kernel time (HIP events on its dispatch) for combinations of nlist and L on the bench
index: (nlist, L) large/small isolates the coarse phase (nlist * M lookups + w + 1 selection rounds) and the candidate phase
(L gathers + L * M lookups) from the fixed part (table build from the 128 KiB codebook, launch)."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from rii_amd import bench_data as bd
dev = torch.device("cuda", 0)
N, B, M = 1_000_000, 1024, 32
base, train, query = bd.sift_like(n_base=N, n_train=100_000, n_query=B)
cw = bd.train_pq(train, M, 256, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
q = torch.from_numpy(np.ascontiguousarray(query[:B])).to(dev)
oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev); oc = torch.empty((B,), dtype=torch.int64, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
out = {}
for nlist in (1024, 64):
    eng.reconfigure(nlist, 2)
    for L in (977, 16):
        def step(): eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2: step(); torch.cuda.synchronize()
        eng.set_option("timing", 2); eng.timing_reset()
        K = 100
        for _ in range(K): step()
        torch.cuda.synchronize()
        ms, n = eng.timing_read("ivf_fused"); eng.set_option("timing", 0)
        out["nlist%d_L%d" % (nlist, L)] = round(ms / n * 1e3, 2)
print(json.dumps(out))
