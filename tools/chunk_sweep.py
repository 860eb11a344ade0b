"""fscan time vs number of chunks at small batches (option scan_chunks)."""
import sys
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
for B in (64, 128, 256, 512, 1024):
    Q = torch.from_numpy(rng.random((B, 128)).astype(np.float32)).to(dev)
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev)
    od = torch.empty((B, 1), dtype=torch.float32, device=dev)
    fn = lambda: g.query_linear_dev(Q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), stream)
    tiles = (B + 15) // 16
    for chunks in (0,) + tuple(c for c in (1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128) if 64 <= tiles * c <= 768):
        g.set_option("scan_chunks", chunks)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g.set_option("timing", 1)
        g.timing_reset()
        for _ in range(5):
            fn()
        ms, n = g.timing_read("scan")
        g.set_option("timing", 0)
        print("B=%4d tiles=%2d chunks=%3d%s blocks=%4d  scan %.1f us   cand_total %d" % (B, tiles, chunks, " (auto)" if chunks == 0 else "", tiles * chunks, 1e3 * ms / n, g.get_option("cand_total")))
