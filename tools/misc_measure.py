"""One-off measurements quoted in DESIGN.md §8 (not part of the bench contract)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from rii_amd import RiiGpu
from tests.util import make_problem

E = np.array([], np.int64)


def t(fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


# config 1: README shape, single query per call (N=10k, D=128, M=32)
cw, codes, qs = make_problem(1, 32, 256, 4, 10000, "sift")
g = RiiGpu(cw, False)
g.add_codes(codes, False)
t0 = time.perf_counter(); g.reconfigure(100, 5); print("reconfigure(100,5) N=10k: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
print("single-query linear  N=10k: %.3f ms/query" % t(lambda: g.query_linear(qs[0], 3, E), 50))
print("single-query ivf L=100 N=10k: %.3f ms/query" % t(lambda: g.query_ivf(qs[0], 3, E, 100), 50))

# 1M codes: reconfigure, ivf top-10, single-query latency
cw, codes, qs = make_problem(2, 32, 256, 4, 1000000, "sift")
Q = np.tile(qs, (64, 1))[:1024] + np.random.default_rng(0).random((1024, 128)).astype(np.float32)
g = RiiGpu(cw, False)
g.add_codes(codes, False)
t0 = time.perf_counter(); g.reconfigure(1024, 5); print("reconfigure(1024,5) N=1M: %.1f ms (reference: 7.2 s on 8 threads)" % ((time.perf_counter() - t0) * 1e3))
L0 = 977
for k in (1, 10, 100):
    print("ivf batch=1024 topk=%d L=%d: %.3f ms/batch" % (k, max(L0, k), t(lambda: g.query_ivf_batch(Q, k, None, max(L0, k)))))
print("ivf batch=1024 topk=10 L=4*L0: %.3f ms/batch" % t(lambda: g.query_ivf_batch(Q, 10, None, 4 * L0)))
print("single-query linear N=1M: %.3f ms/query" % t(lambda: g.query_linear(qs[0], 1, E), 20))
print("single-query ivf    N=1M: %.3f ms/query" % t(lambda: g.query_ivf(qs[0], 1, E, L0), 20))
sub = np.sort(np.random.default_rng(1).choice(1000000, 100000, replace=False)).astype(np.int64)
print("ivf batch=1024 topk=1 S=100k: %.3f ms/batch" % t(lambda: g.query_ivf_batch(Q, 1, sub, L0)))
print("linear batch=1024 topk=1 (host pointers, PCIe-inclusive): %.3f ms/batch" % t(lambda: g.query_linear_batch(Q, 1, None)))
new = make_problem(3, 32, 256, 4, 100000, "sift")[1]
t0 = time.perf_counter(); g.add_codes(new, True); print("add_codes(100k, update=True): %.1f ms" % ((time.perf_counter() - t0) * 1e3))

# the reference's own SIFT1M benchmark shape: M=64 (examples/benchmark/run_sift1m.py:60-61) -> exact scan, QT=2
cw, codes, qs = make_problem(4, 64, 256, 2, 1000000, "unit")
Q = np.tile(qs, (64, 1))[:1024].copy()
g = RiiGpu(cw, False)
g.add_codes(codes, False)
print("linear batch=1024 topk=1 M=64: %.3f ms/batch" % t(lambda: g.query_linear_batch(Q, 1, None), 3))
