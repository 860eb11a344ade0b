import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from rii_amd import RiiGpu
rng = np.random.default_rng(1)
M, Ks, N, B = 32, 256, 1000000, 1024
cw = (rng.random((M, Ks, 4)) * 255).astype(np.float32)
codes = rng.integers(0, Ks, size=(N, M), dtype=np.uint8)
Q = (rng.random((B, M * 4)) * 255).astype(np.float32)
g = RiiGpu(cw, False)
g.add_codes(codes, False)
q = torch.from_numpy(Q).cuda()
ids = torch.empty((B, 1), dtype=torch.int64, device="cuda"); d = torch.empty((B, 1), dtype=torch.float32, device="cuda")
for timing in (0, 1, 0, 1):
    g.set_option("timing", timing); g.timing_reset()
    for _ in range(3):
        g.query_linear_dev(q.data_ptr(), B, 1, 0, 0, ids.data_ptr(), d.data_ptr())
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.query_linear_dev(q.data_ptr(), B, 1, 0, 0, ids.data_ptr(), d.data_ptr())
    g.synchronize()
    print("timing=%d step %.4f ms" % (timing, (time.perf_counter() - t0) / 50 * 1e3))
