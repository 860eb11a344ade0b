#!/usr/bin/env python3
"""Round 6: the coarse pre-pass of the database-sharded inverted index (shard_coarse_quad_kernel + ivf_shard_any_kernel<PRE>) against the
walk kernel doing its own coarse phase, kernel times by HIP events through rii_query_ivf_shard_dev (one rank: G = 1), Deep1B shard shape
(M = 16, Ds = 6, uniform random codes, modulo partition, nlist = L = sqrt-ish) and the SIFT shape; rows compared bit for bit."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rii_amd import RiiGpu
dev = torch.device("cuda", 0)
out = {}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
SHAPES = ((16, 6, N, 8000, 8000), (32, 4, 1_000_000, 1024, 977))
for M, Ds, n, NL, L in SHAPES:
    rng = np.random.default_rng(1)
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    codes = torch.randint(0, 256, (n, M), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).numpy()
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
    del codes
    cen = rng.integers(0, 256, size=(NL, M), dtype=np.uint8)
    off, ids = bench.modulo_lists(n, NL)
    eng.set_posting_lists(cen, off, ids)
    glen = torch.from_numpy(np.diff(off).astype(np.int32)).to(dev)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    for B in (1024, 128, 16):
        q = torch.from_numpy(rng.random((B, M * Ds)).astype(np.float32)).to(dev)
        res, rows = {}, {}
        for pre in (0, 2):
            eng.set_option("shard_pre", pre)
            oi = torch.empty((B, 2), dtype=torch.int64, device=dev); od = torch.empty((B, 2), dtype=torch.float32, device=dev)
            op = torch.empty((B, 2), dtype=torch.int32, device=dev); on = torch.empty((B,), dtype=torch.int32, device=dev)
            oc = torch.empty((B,), dtype=torch.int64, device=dev)
            def step(): eng.query_ivf_shard_dev(q.data_ptr(), B, 1, 0, 0, 0, L, n, glen.data_ptr(), 1, 0, oi.data_ptr(), od.data_ptr(), op.data_ptr(),
                                                on.data_ptr(), oc.data_ptr(), st.cuda_stream)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.1: step(); torch.cuda.synchronize()
            eng.set_option("timing", 2); eng.timing_reset()
            for _ in range(50): step()
            torch.cuda.synchronize()
            ms, k = eng.timing_read("ivf_shard"); ms2, k2 = eng.timing_read("shard_coarse"); eng.set_option("timing", 0)
            t0 = time.perf_counter()
            for _ in range(50): step()
            torch.cuda.synchronize()
            res["pre%d" % pre] = {"walk_kernel_us": round(ms / max(k, 1) * 1e3, 2), "coarse_kernel_us": round(ms2 / max(k2, 1) * 1e3, 2) if k2 else 0.0,
                                  "step_us_no_events": round((time.perf_counter() - t0) / 50 * 1e6, 2)}
            rows[pre] = (oi.cpu().numpy().copy(), od.cpu().numpy().copy(), op.cpu().numpy().copy())
        if B == 1024:                                      # the pre-pass cut short after its table (11) / scoring (12) phase
            eng.set_option("shard_pre", 2)
            for stop in (11, 12):
                eng.set_option("shard_dbg_stop", stop)
                for _ in range(5): step()
                torch.cuda.synchronize()
                eng.set_option("timing", 2); eng.timing_reset()
                for _ in range(50): step()
                torch.cuda.synchronize()
                ms2, k2 = eng.timing_read("shard_coarse"); eng.set_option("timing", 0)
                res["coarse_stop%d_us" % stop] = round(ms2 / max(k2, 1) * 1e3, 2)
            for stop in (1, 3, 4, 5):                      # the walk kernel behind the pre-pass, cut short after its table load / order / walk / candidates
                eng.set_option("shard_dbg_stop", stop)
                for _ in range(5): step()
                torch.cuda.synchronize()
                eng.set_option("timing", 2); eng.timing_reset()
                for _ in range(50): step()
                torch.cuda.synchronize()
                ms2, k2 = eng.timing_read("ivf_shard"); eng.set_option("timing", 0)
                res["walk_stop%d_us" % stop] = round(ms2 / max(k2, 1) * 1e3, 2)
            eng.set_option("shard_dbg_stop", 0)
        res["rows_equal"] = bool(all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(rows[0], rows[2])))
        out["M%d_Ds%d_N%d_nlist%d_L%d_B%d" % (M, Ds, n, NL, L, B)] = res
    eng.set_option("shard_pre", 1)
    del eng
print(json.dumps(out))
