"""Per-kernel time of the linear top-1 path at mid-size batches (device-resident inputs)."""
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
g.set_option("fast_min_batch", 0)
for B in (32, 64, 128, 256, 512, 1024):
    Q = torch.from_numpy(rng.random((B, 128)).astype(np.float32)).to(dev)
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev)
    od = torch.empty((B, 1), dtype=torch.float32, device=dev)
    fn = lambda: g.query_linear_dev(Q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), stream)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g.set_option("timing", 1)
    g.timing_reset()
    for _ in range(10):
        fn()
    parts = {k: g.timing_read(k) for k in ("lut", "quant", "scan", "rerank")}
    g.set_option("timing", 0)
    print("B=%4d  " % B + "  ".join("%s %.1f us" % (k, 1e3 * v[0] / max(v[1], 1)) for k, v in parts.items()),
          " cand_total %d cand_max %d" % (g.get_option("cand_total"), g.get_option("cand_max")))
