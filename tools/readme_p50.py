#!/usr/bin/env python3
"""p50 of the README one-query calls (N = 10k, nlist = 100, L = 100): tools/readme_p50.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rii_amd import RiiGpu
from tests.util import make_problem
E = np.array([], np.int64)
cw, codes, qs = make_problem(1, 32, 256, 4, 10000, "unit")
g = RiiGpu(cw, False); g.add_codes(codes, False); g.reconfigure(100, 5)
out = {}
for rep in range(2):
    for name, fn in (("ivf_top3", lambda q: g.query_ivf(q, 3, E, 100)), ("ivf_top1", lambda q: g.query_ivf(q, 1, E, 100)), ("linear_top3", lambda q: g.query_linear(q, 3, E))):
        for i in range(200): fn(qs[i % 16])
        ts = []
        for i in range(600):
            t0 = time.perf_counter(); fn(qs[i % 16]); ts.append(time.perf_counter() - t0)
        out.setdefault(name, []).append(round(float(np.percentile(np.array(ts) * 1e6, 50)), 2))
print(out)
