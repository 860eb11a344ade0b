import sys, time
import numpy as np
sys.path.insert(0, ".")
import rii_amd
from rii_amd.codec import PQ
from tests.util import make_problem
rng = np.random.default_rng(0)
X = rng.random((200000, 128)).astype(np.float32)
codec = PQ(M=32, Ks=256, verbose=False).fit(X[:2000], iter=3)
e = rii_amd.Rii(codec)
t0 = time.perf_counter(); e.add(X, update_posting_lists=False); print("add 200k (host encode + upload): %.2f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); e.impl_cpp.reconfigure(447, 5); print("impl.reconfigure(447,5): %.3f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); e.reconfigure(447, 5); print("Rii.reconfigure incl. threshold estimation: %.3f s" % (time.perf_counter() - t0))
print("threshold:", e.threshold)
q = X[0]
for m in ("linear", "ivf", "auto"):
    t0 = time.perf_counter()
    for _ in range(200):
        e.query(q, topk=3, method=m)
    print("Rii.query method=%s: %.3f ms/query" % (m, (time.perf_counter() - t0) / 200 * 1e3))
S = np.sort(rng.choice(200000, 5000, replace=False)).astype(np.int64)
t0 = time.perf_counter()
for _ in range(200):
    e.query(q, topk=3, target_ids=S)
print("Rii.query |S|=5000 auto: %.3f ms/query" % ((time.perf_counter() - t0) / 200 * 1e3))
