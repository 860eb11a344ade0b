#!/usr/bin/env python3
"""Round 6: the headline linear path on TIGHTER SIFT-shaped clusters than bench.py's default set (noise sigma 24 -> 12 / 6, fewer
clusters): filter candidates per query and step time at the fused tables' 63 / 127 / 255 levels -- does the headline share the
generic quantiser's cliff (profiles/r06_deep_structured_levels.json)?"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rii_amd import RiiGpu, bench_data as bd
dev = torch.device("cuda", 0)
N, D, M, B = 1_000_000, 128, 32, 1024
out = []
for sigma, ncl in ((24.0, 4096), (12.0, 4096), (6.0, 1024), (3.0, 256)):
    rng = np.random.default_rng(5)
    means = (rng.random((ncl, D), dtype=np.float32) * 128.0)
    def draw(n, seed):
        r = np.random.default_rng(seed)
        c = r.integers(0, ncl, n)
        return np.rint(np.clip(means[c] + r.standard_normal((n, D), dtype=np.float32) * sigma, 0, 255)).astype(np.float32)
    base, train, query = draw(N, 1), draw(100_000, 2), draw(B, 3)
    cw = bd.train_pq(train, M, 256, iters=8, seed=123, device=dev)
    codes = bd.encode_pq(base, cw, device=dev)
    gt = bd.exact_nn(base, query, device=dev)
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
    q = torch.from_numpy(query).to(dev)
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for lv in (63, 127, 255):
        eng.set_option("table_levels", lv)
        def step(): eng.query_linear_dev(q.data_ptr(), B, 1, 0, 0, oi.data_ptr(), od.data_ptr(), st)
        for _ in range(30): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        eng.set_option("fused_rerank", 0)
        step(); torch.cuda.synchronize()
        cand = eng.get_option("cand_total") / B; cmax = eng.get_option("cand_max")
        eng.set_option("fused_rerank", 1)
        out.append({"sigma": sigma, "clusters": ncl, "levels": lv, "ms_per_step": round(ms, 4), "candidates_per_query": round(cand, 1), "longest_list": cmax,
                    "recall_at_1": bd.recall_at_r(oi.cpu().numpy(), gt, 1)})
        print(json.dumps(out[-1]), flush=True)
    del eng
