import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from rii_amd import RiiGpu
from rii_amd import dist as rd
rng = np.random.default_rng(3)
N, M, nlist, B = 300_000, 32, 1024, 9001
cw = rng.random((M, 256, 4)).astype(np.float32)
codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
g = RiiGpu(cw, False, device=0); g.add_codes(codes, False)
off, ids = bench.modulo_lists(N, nlist)
g.set_posting_lists(rng.integers(0, 256, size=(nlist, M), dtype=np.uint8), off, ids)
Q = rng.random((B, M * 4)).astype(np.float32)
E = np.array([], np.int64)
L = N // nlist
g.set_option("ivf_quad", 0); a = g.query_ivf_batch(Q, 1, E, L)
g.set_option("ivf_quad", 1); b = g.query_ivf_batch(Q, 1, E, L)
g.set_option("ivf_quad", 2); c = g.query_ivf_batch(Q[:777], 1, E, L)
assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
assert np.array_equal(a[0][:777], c[0]) and np.array_equal(a[1][:777].view(np.uint32), c[1].view(np.uint32))
idx = rd.DbShardedIndex(g, 0, N)
for topk, LL in ((1, L), (1, 20000), (4, L)):
    want = g.query_ivf_batch(Q, topk, E, LL)
    gi, gd, gc = idx.query_ivf_batch(torch.from_numpy(Q).cuda(), topk, None, LL)
    assert np.array_equal(gi.cpu().numpy(), want[0]) and np.array_equal(gd.cpu().numpy().view(np.uint32), want[1].view(np.uint32)) and np.array_equal(gc.cpu().numpy(), want[2]), (topk, LL)
li = g.query_linear_batch(Q, 1, E)
gi, gd = idx.query_linear_batch(torch.from_numpy(Q).cuda(), 1)
assert np.array_equal(gi.cpu().numpy(), li[0])
print("big-batch checks OK")
