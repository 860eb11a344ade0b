"""QPS vs batch size (device-resident inputs) for the linear and inverted-index paths; quoted in DESIGN.md."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from rii_amd import RiiGpu
from tests.util import make_problem

cw, codes, qs = make_problem(2, 32, 256, 4, 1000000, "unit")
g = RiiGpu(cw, False)
g.add_codes(codes, False)
g.reconfigure(1024, 2)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
stream = torch.cuda.current_stream().cuda_stream
for B in (1, 4, 16, 64, 256, 1024, 4096, 8192):
    Q = torch.from_numpy(rng.random((B, 128)).astype(np.float32)).to(dev)
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev)
    od = torch.empty((B, 1), dtype=torch.float32, device=dev)
    oc = torch.empty((B,), dtype=torch.int64, device=dev)
    res = {}
    for name, fn in (("linear", lambda: g.query_linear_dev(Q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), stream)),
                     ("ivf", lambda: g.query_ivf_dev(Q.data_ptr(), B, 1, 0, 0, 977, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), stream))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[name] = (dt * 1e3, B / dt)
    print("B=%5d  linear %.3f ms (%.0f QPS)   ivf %.3f ms (%.0f QPS)" % (B, res["linear"][0], res["linear"][1], res["ivf"][0], res["ivf"][1]))
