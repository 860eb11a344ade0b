"""The reference README's usage (README.md:80-140 there), unchanged except for the import line and the codec
(nanopq is replaced by the stand-in of rii_amd.codec when it is not installed).  Needs an MI355X."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run from a checkout

try:
    import nanopq
    PQ = nanopq.PQ
except ImportError:                       # no network in the build image
    from rii_amd.codec import PQ

import rii_amd as rii                     # instead of: import rii

N, Nt, D = 10000, 1000, 128
X = np.random.random((N, D)).astype(np.float32)      # 10,000 128-dim vectors to be searched
Xt = np.random.random((Nt, D)).astype(np.float32)    # 1,000 128-dim vectors for training
q = np.random.random((D,)).astype(np.float32)        # a 128-dim vector

codec = PQ(M=32, Ks=256, verbose=False).fit(vecs=Xt)
e = rii.Rii(fine_quantizer=codec)
e.add_configure(vecs=X)

ids, dists = e.query(q=q, topk=3)
print(ids, dists)

S = np.array([2, 24, 43, 55, 102, 139, 221, 542, 667, 873, 874, 899], dtype=np.int64)
ids, dists = e.query(q=q, topk=3, target_ids=S)      # subset search
print(ids, dists)

t0 = time.time()
for _ in range(100):
    e.query(q=q, topk=3)
print((time.time() - t0) * 10, "msec/query (one query per call)")

Q = np.random.random((1024, D)).astype(np.float32)
t0 = time.time()
ids, dists, counts = e.query_batch(Q, topk=3)        # NEW: a whole batch per call
print((time.time() - t0) * 1e3 / 1024, "msec/query (batch of 1024)")

X2 = np.random.random((1000, D)).astype(np.float32)
e.add(vecs=X2)                                       # posting lists are updated on the GPU
e.reconfigure(nlist=200)
print(e.N, e.nlist, e.L0)
