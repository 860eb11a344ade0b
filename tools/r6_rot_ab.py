#!/usr/bin/env python3
"""Round 6: ivf_rot_kernel against ivf_fused_kernel, kernel time by HIP events at B = 1024 over a 1M-code index (random codes, modulo
partition: the phases' cost does not depend on what the lists hold), rows compared."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rii_amd import RiiGpu
dev = torch.device("cuda", 0)
N, B = 1_000_000, 1024
out = {}
SHAPES = ((64, 2, 1000, (5000, 2048, 16000, 16)), (64, 2, 16, (16, 5000)), (32, 4, 1024, (977, 5000, 16000)), (64, 4, 256, (5000,)))
if len(sys.argv) > 1 and sys.argv[1] == "m64":
    SHAPES = ((64, 2, 1000, (5000, 16)), (64, 2, 16, (16, 5000)))
for M, Ds, NL, Ls in SHAPES:
    rng = np.random.default_rng(1)
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
    eng.set_option("ivf_quad", 0)
    q = torch.from_numpy(rng.random((B, M * Ds)).astype(np.float32)).to(dev)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    cen = rng.integers(0, 256, size=(NL, M), dtype=np.uint8)
    off, ids = bench.modulo_lists(N, NL)
    eng.set_posting_lists(cen, off, ids)
    for L in Ls:
        res = {}
        rows = {}
        for rot in (0, 2):
            eng.set_option("ivf_rot", rot)
            n0 = eng.get_option("ivf_rot_launches")
            oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev); oc = torch.empty((B,), dtype=torch.int64, device=dev)
            def step(): eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.1: step(); torch.cuda.synchronize()
            eng.set_option("timing", 2); eng.timing_reset()
            for _ in range(100): step()
            torch.cuda.synchronize()
            ms, n = eng.timing_read("ivf_fused"); eng.set_option("timing", 0)
            res["rot%d_us" % rot] = round(ms / max(n, 1) * 1e3, 2)
            res["rot%d_used_rot_kernel" % rot] = eng.get_option("ivf_rot_launches") > n0
            rows[rot] = (oi.cpu().numpy().copy(), od.cpu().numpy().copy())
        res["rows_equal"] = bool(np.array_equal(rows[0][0], rows[2][0]) and np.array_equal(rows[0][1].view(np.uint32), rows[2][1].view(np.uint32)))
        res["speedup"] = round(res["rot0_us"] / res["rot2_us"], 3)
        out["M%d_Ds%d_nlist%d_L%d" % (M, Ds, NL, L)] = res
    del eng
print(json.dumps(out))
if len(sys.argv) > 1 and sys.argv[1] == "m64":
    # the rot kernel cut short after its table / coarse / selection phase (option ivf_dbg_stop)
    M, Ds, NL, L = 64, 2, 1000, 5000
    rng = np.random.default_rng(1)
    cw = rng.random((M, 256, Ds)).astype(np.float32)
    codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
    eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
    eng.set_option("ivf_quad", 0); eng.set_option("ivf_rot", 2)
    q = torch.from_numpy(rng.random((B, M * Ds)).astype(np.float32)).to(dev)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st)
    cen = rng.integers(0, 256, size=(NL, M), dtype=np.uint8)
    off, ids = bench.modulo_lists(N, NL)
    eng.set_posting_lists(cen, off, ids)
    oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev); oc = torch.empty((B,), dtype=torch.int64, device=dev)
    ph = {}
    for stop in (9, 1, 2, 3, 0):
        eng.set_option("ivf_dbg_stop", stop)
        def step(): eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1: step(); torch.cuda.synchronize()
        eng.set_option("timing", 2); eng.timing_reset()
        for _ in range(100): step()
        torch.cuda.synchronize()
        ms, n = eng.timing_read("ivf_fused"); eng.set_option("timing", 0)
        ph["stop%d_us" % stop] = round(ms / max(n, 1) * 1e3, 2)
    eng.set_option("ivf_dbg_stop", 0)
    print(json.dumps({"rot_kernel_phases_M64_nlist1000_L5000": ph}))
