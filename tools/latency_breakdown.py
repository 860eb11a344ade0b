import sys, time, ctypes
import numpy as np
sys.path.insert(0, ".")
from rii_amd import RiiGpu, core
from tests.util import make_problem
E = np.array([], np.int64)
cw, codes, qs = make_problem(1, 32, 256, 4, 10000, "unit")
g = RiiGpu(cw, False); g.add_codes(codes, False); g.reconfigure(100, 5)
L = core._lib()
q = np.ascontiguousarray(qs[0]); ids = np.empty(3, np.int64); d = np.empty(3, np.float32); cnt = np.empty(1, np.int64)
P = core._ptr
def t(fn, n=300):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6
print("C call rii_query_ivf    topk=3 L=100: %.1f us" % t(lambda: L.rii_query_ivf(g._h, P(q, ctypes.c_float), 1, 3, P(E, ctypes.c_int64), 0, 100, P(ids, ctypes.c_int64), P(d, ctypes.c_float), P(cnt, ctypes.c_int64))))
print("C call rii_query_linear topk=3      : %.1f us" % t(lambda: L.rii_query_linear(g._h, P(q, ctypes.c_float), 1, 3, P(E, ctypes.c_int64), 0, P(ids, ctypes.c_int64), P(d, ctypes.c_float))))
print("C call rii_query_linear topk=1      : %.1f us" % t(lambda: L.rii_query_linear(g._h, P(q, ctypes.c_float), 1, 1, P(E, ctypes.c_int64), 0, P(ids, ctypes.c_int64), P(d, ctypes.c_float))))
print("RiiGpu.query_ivf  (python wrapper)  : %.1f us" % t(lambda: g.query_ivf(q, 3, E, 100)))
print("RiiGpu.query_linear topk=1 (python) : %.1f us" % t(lambda: g.query_linear(q, 1, E)))
g.set_option("timing", 1)
for _ in range(50): g.query_ivf(q, 3, E, 100)
for k in ("lut", "ivf_fused", "ivf_plan", "ivf_scan", "ivf_select"):
    ms, n = g.timing_read(k); print("  kernel %s: %.1f us x %d" % (k, ms / max(n, 1) * 1e3, n))
