#!/usr/bin/env python3
"""Round 5: same-box A/B of two builds of the library on the filter scans (tools/ab_so.sh-style: the caller swaps rii_amd/librii_amd.so).
Random codes and codebooks (the scan's cost does not depend on the data beyond the candidate rate, which random data keep low);
kernel time from the events attached to the scan's dispatch (timing = 2) and the un-instrumented step."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
out = {}
for M, Ds, N, Bs in ((32, 4, 1_000_000, (1024, 128)), (16, 6, 16_000_000, (1024,))):
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for B in Bs:
        q = torch.from_numpy(rng.random((B, M * Ds)).astype(np.float32)).to(dev)
        oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev)
        def step(): eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), st.cuda_stream)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3: step(); torch.cuda.synchronize()
        K = 60 if N <= 1_000_000 else 12
        best = None
        for rep in range(3):
            eng.set_option("timing", 2); eng.timing_reset()
            for _ in range(K): step()
            torch.cuda.synchronize()
            ms, n = eng.timing_read("scan"); eng.set_option("timing", 0)
            t0 = time.perf_counter()
            for _ in range(K): step()
            torch.cuda.synchronize()
            r = (round(ms / n * 1e3, 2), round((time.perf_counter() - t0) / K * 1e6, 2))
            best = r if best is None or r[0] < best[0] else best
        out["M%d_N%d_B%d" % (M, N, B)] = best
    del eng
print(json.dumps(out))
