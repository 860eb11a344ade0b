"""Probe: how much of fscan's time is LDS bank conflicts, and which lanes share a ds_read_b128 service group?
Arrange codes so that the 16 lanes of a (hypothesised) service group read identical table rows (broadcast, no conflict)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rii_amd import RiiGpu
from tests.util import make_problem

N, M = 1 << 20, 32
cw, _, qs = make_problem(1, M, 256, 4, 16, "unit")
rng = np.random.default_rng(0)
Q = rng.random((1024, 128)).astype(np.float32)
GUIDE = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
         list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
         list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
         list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
CONTIG = [list(range(16 * g, 16 * g + 16)) for g in range(4)]


def arrange(groups, distinct_nibbles=False):
    codes = np.empty((N, M), np.uint8)
    blocks = N // 64
    if not distinct_nibbles:         # every lane of a group holds the same code -> pure broadcast
        g4 = rng.integers(0, 256, size=(blocks, 4, M), dtype=np.uint8)
        for gi, lanes in enumerate(groups):
            for l in lanes:
                codes.reshape(blocks, 64, M)[:, l, :] = g4[:, gi, :]
    else:                            # within a group all 16 low nibbles differ (per m) -> conflict-free, no broadcast
        hi = rng.integers(0, 16, size=(blocks, 64, M), dtype=np.uint8)
        for gi, lanes in enumerate(groups):
            perm = np.argsort(rng.random((blocks, M, 16)), axis=2).astype(np.uint8)   # a permutation of 0..15 per (block, m)
            for j, l in enumerate(lanes):
                codes.reshape(blocks, 64, M)[:, l, :] = (hi[:, l, :] << 4) | perm[:, :, j]
    return codes


def run(name, codes):
    g = RiiGpu(cw, False)
    g.add_codes(codes, False)
    g.set_option("timing", 1)
    for mode in (1, 0):
        g.set_option("scan_mode", mode)
        g.query_linear_batch(Q, 1, None)
        g.timing_reset()
        for _ in range(5):
            g.query_linear_batch(Q, 1, None)
        ms, n = g.timing_read("scan")
        print("%-34s scan_mode=%d  %.3f ms per launch" % (name, mode, ms / n))


run("random codes", rng.integers(0, 256, size=(N, M), dtype=np.uint8))
run("guide groups, distinct nibbles", arrange(GUIDE, True))
run("contiguous-16, distinct nibbles", arrange(CONTIG, True))
run("guide groups, broadcast", arrange(GUIDE, False))
