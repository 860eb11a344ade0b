"""Measure the LDS-friendly scan order: build time of the permutation and the filter kernel with / without it."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rii_amd import RiiGpu

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = np.random.default_rng(0)
cw = rng.random((M, 256, 4)).astype(np.float32)
codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
Q = rng.random((1024, M * 4)).astype(np.float32)
g = RiiGpu(cw, False)
g.add_codes(codes, False)
g.set_option("timing", 1)
for order in (0, 1, 0, 1):
    g.set_option("scan_order", order)
    g.query_linear_batch(Q, 1, None)
    if order:
        ms, n = g.timing_read("scan_order")
        if n:
            print("scan_order build: %.3f ms for %d codes (%d launches)" % (ms, N, n))
    g.timing_reset()
    t0 = time.perf_counter()
    for _ in range(10):
        ids, d = g.query_linear_batch(Q, 1, None)
    wall = (time.perf_counter() - t0) / 10
    ms, n = g.timing_read("scan")
    print("scan_order=%d  fscan %.4f ms/launch   host call %.3f ms   checksum %d" % (order, ms / n, wall * 1e3, int(ids.sum())))
    g.timing_reset()
for k in (10, 100):
    for order in (0, 1):
        g.set_option("scan_order", order)
        g.query_linear_batch(Q, k, None)
        g.timing_reset()
        for _ in range(5):
            ids, d = g.query_linear_batch(Q, k, None)
        ms, n = g.timing_read("scan")
        print("topk=%d scan_order=%d  fscan %.4f ms/launch (%d launches)" % (k, order, ms / n, n))
