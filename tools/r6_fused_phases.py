#!/usr/bin/env python3
"""Round 6: where ivf_fused_kernel's time goes at the reference's own harness shape (M = 64, Ds = 2, nlist = 1000, L = 5000, B = 1024)
and at the Deep-shaped one (M = 16, Ds = 6): kernel time by HIP events for (nlist, L) large and small -- the differences are the
candidate phase, the coarse phase, and what is left (launch, table build, selection, output)."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rii_amd import RiiGpu
dev = torch.device("cuda", 0)
N, B = 1_000_000, 1024
out = {}
for M, Ds, NL, LL in ((64, 2, 1000, 5000), (32, 4, 1024, 977), (16, 6, 1000, 8000)):
    rng = np.random.default_rng(1)
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
    eng.set_option("ivf_quad", 0)
    q = torch.from_numpy(rng.random((B, M * Ds)).astype(np.float32)).to(dev)
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev); oc = torch.empty((B,), dtype=torch.int64, device=dev)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    def kernel_us(L, K=100):
        def step(): eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1: step(); torch.cuda.synchronize()
        eng.set_option("timing", 2); eng.timing_reset()
        for _ in range(K): step()
        torch.cuda.synchronize()
        ms, n = eng.timing_read("ivf_fused"); eng.set_option("timing", 0)
        return round(ms / max(n, 1) * 1e3, 2)
    res = {}
    for nlist in (NL, 16):
        cen = rng.integers(0, 256, size=(nlist, M), dtype=np.uint8)
        off, ids = bench.modulo_lists(N, nlist)
        eng.set_posting_lists(cen, off, ids)
        for L in (LL, 16):
            res["nlist%d_L%d" % (nlist, L)] = kernel_us(L)
    res["candidates_us"] = round(res["nlist%d_L%d" % (NL, LL)] - res["nlist%d_L16" % NL], 2)
    res["coarse_us"] = round(res["nlist%d_L16" % NL] - res["nlist16_L16"], 2)
    res["rest_us"] = res["nlist16_L16"]
    out["M%d_Ds%d" % (M, Ds)] = res
    del eng
print(json.dumps(out))
