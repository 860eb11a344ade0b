import sys, numpy as np, torch, time
sys.path.insert(0, ".")
from rii_amd import RiiGpu, host_simd_arch
from rii_amd import bench_data as bd
dev = torch.device("cuda", 0)
N, M, Ks, D, B = 1000000, 32, 256, 128, 1024
base, train, query = bd.sift_like(n_base=N, n_train=100000, n_query=10000, D=D)
cw = bd.train_pq(train, M, Ks, iters=10, seed=123, device=dev)
codes = bd.encode_pq(base, cw, device=dev)
g = RiiGpu(cw, False)
g.add_codes(codes, False)
g.reconfigure(1024, 5)
Q = query[:B]
for k in (10, 100):
    ids, d, cnt = g.query_ivf_batch(Q, k, None, 977)
    tie = (np.diff(d, axis=1) == 0).any(axis=1)
    print("k=%d rows with an exact tie inside the top-k: %d of %d" % (k, tie.sum(), B))
    g.set_option("timing", 1); g.timing_reset()
    t0 = time.perf_counter(); g.query_ivf_batch(Q, k, None, 977); el = time.perf_counter() - t0
    print("  call %.3f ms  fused %s exact %s" % (el * 1e3, g.timing_read("ivf_fused"), g.timing_read("ivf_exact")))
    g.set_option("timing", 0)
u = np.unique(codes, axis=0).shape[0]
print("distinct codes:", u, "of", N)
