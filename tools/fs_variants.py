#!/usr/bin/env python3
"""Quick A/B harness for fscan_kernel experiments: times the filter scan on the bench shape (N=1M, M=32, B=1024, random
codes: no codec training, ~6 s per run on the GPU box).  Each argument is a value of the environment variable
RII_FS_VARIANT handed to a fresh process (an experimental build reads it in launch_fscan_mode; the committed code
ignores it):   python tools/fs_variants.py 0 1   -> kernel ms (HIP events), step ms, candidates per query, ids checksum."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
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
for _ in range(3):
    g.query_linear_dev(q.data_ptr(), B, 1, 0, 0, ids.data_ptr(), d.data_ptr())
g.synchronize()
g.set_option("timing", 1); g.timing_reset()
t0 = time.perf_counter()
for _ in range(20):
    g.query_linear_dev(q.data_ptr(), B, 1, 0, 0, ids.data_ptr(), d.data_ptr())
g.synchronize()
el = (time.perf_counter() - t0) / 20 * 1e3
ms, n = g.timing_read("scan")
print("variant %%s: fscan %%.4f ms, step %%.4f ms, cand/query %%.0f, ids checksum %%d" %% (sys.argv[1], ms / n, el, g.get_option("cand_total") / B, int(ids.sum().item())))
""" % ROOT

for v in sys.argv[1:] or ["0"]:
    env = dict(os.environ, RII_FS_VARIANT=v)
    subprocess.run([sys.executable, "-c", WORKER, v], env=env, check=False)
