"""Where does the byte-table filter start to beat the exhaustive fp32 scan?  (engine option fast_min_batch)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rii_amd import RiiGpu
from tests.util import make_problem

cw, codes, qs = make_problem(2, 32, 256, 4, 1000000, "unit")
g = RiiGpu(cw, False)
g.add_codes(codes, False)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
stream = torch.cuda.current_stream().cuda_stream
for B in (2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256):
    Q = torch.from_numpy(rng.random((B, 128)).astype(np.float32)).to(dev)
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev)
    od = torch.empty((B, 1), dtype=torch.float32, device=dev)
    out = []
    for fmb in (1 << 30, 0):          # exhaustive scan / filter + re-rank
        g.set_option("fast_min_batch", fmb)
        fn = lambda: g.query_linear_dev(Q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), stream)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    print("B=%4d  exhaustive %.3f ms   filter %.3f ms" % (B, out[0], out[1]))
