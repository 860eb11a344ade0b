"""Cost of the exact std::partial_sort replay (ivf_exact_lds_kernel) per query: every query forced through it."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from rii_amd import RiiGpu
from tests.util import make_problem
cw, codes, qs = make_problem(2, 32, 256, 4, 1000000, "unit")
g = RiiGpu(cw, False)
g.add_codes(codes, False)
g.reconfigure(1024, 2)
rng = np.random.default_rng(0)
Q = rng.random((1024, 128)).astype(np.float32)
for B in (1, 256, 1024):
    for k in (1, 10, 100):
        for force in (0, 1):
            g.set_option("ivf_force_exact", force)
            g.query_ivf_batch(Q[:B], k, None, 977)
            g.set_option("timing", 1); g.timing_reset()
            g.query_ivf_batch(Q[:B], k, None, 977)
            f, e = g.timing_read("ivf_fused"), g.timing_read("ivf_exact")
            g.set_option("timing", 0)
            print("B=%4d k=%3d force=%d  fused %.3f ms  exact %.3f ms" % (B, k, force, f[0], e[0]))
