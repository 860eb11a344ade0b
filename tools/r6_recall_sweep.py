import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rii_amd import bench_data as bd, RiiGpu
dev = torch.device("cuda", 0)
n, D, M = 4_000_000, 96, 16
for sigma, decay, ncl in ((0.45, 0.7, 16384), (0.45, 1.0, 16384), (0.8, 0.7, 16384), (0.8, 1.0, 4096), (1.5, 1.0, 4096), (0.45, 0.4, 16384)):
    base = bd.deep_like_torch(n, D, seed=77, device=dev, sigma=sigma, decay=decay, n_clusters=ncl)
    train = bd.deep_like_torch(100_000, D, seed=77, device=dev, stream=1, sigma=sigma, decay=decay, n_clusters=ncl)
    query = bd.deep_like_torch(1024, D, seed=77, device=dev, stream=2, sigma=sigma, decay=decay, n_clusters=ncl)
    cw = bd.train_pq(train.cpu().numpy(), M, 256, iters=8, seed=123, device=dev)
    codes = bd.encode_pq_torch(base, cw).cpu().numpy()
    gt = bd.exact_nn_torch(base, query)
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
    ids, d = eng.query_linear_batch(query.cpu().numpy(), 10, np.array([], np.int64))
    print(json.dumps({"sigma": sigma, "decay": decay, "ncl": ncl, "r1": bd.recall_at_r(ids, gt, 1), "r10": bd.recall_at_r(ids, gt, 10)}), flush=True)
    del eng, base
