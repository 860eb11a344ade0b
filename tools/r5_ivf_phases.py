#!/usr/bin/env python3
"""Round 5: where the time of the inverted-index step goes, one query per block (ivf_fused_kernel) against four (ivf_quad_kernel).
Kernel time from HIP events on the dispatch (engine option timing = 2) at B = 1024 over a 1M-code index with random codes and a
modulo partition (the phases' cost does not depend on what the lists hold): the quad kernel cut short after its table / coarse /
selection phase (option ivf_dbg_stop), and both kernels for (nlist, L) large and small."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rii_amd import RiiGpu
dev = torch.device("cuda", 0)
N, B, M = 1_000_000, 1024, int(os.environ.get("M", "32"))
rng = np.random.default_rng(1)
cw = rng.random((M, 256, 4)).astype(np.float32)
codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
q = torch.from_numpy(rng.random((B, M * 4)).astype(np.float32)).to(dev)
oi = torch.empty((B, 1), dtype=torch.int64, device=dev); od = torch.empty((B, 1), dtype=torch.float32, device=dev); oc = torch.empty((B,), dtype=torch.int64, device=dev)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
out = {}
def kernel_us(L, K=200):
    def step(): eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15: step(); torch.cuda.synchronize()
    eng.set_option("timing", 2); eng.timing_reset()
    for _ in range(K): step()
    torch.cuda.synchronize()
    ms, n = eng.timing_read("ivf_fused"); eng.set_option("timing", 0)
    t0 = time.perf_counter()
    for _ in range(K): step()
    torch.cuda.synchronize()
    return round(ms / n * 1e3, 2), round((time.perf_counter() - t0) / K * 1e6, 2)
for nlist in (1024, 64):
    cen = rng.integers(0, 256, size=(nlist, M), dtype=np.uint8)
    off, ids = bench.modulo_lists(N, nlist)
    eng.set_posting_lists(cen, off, ids)
    for L in (977, 16):
        for quad in (1, 0):
            eng.set_option("ivf_quad", quad)
            for stop in ((0, 1, 2, 3) if (quad and nlist == 1024 and L == 977) else (0,)):
                eng.set_option("ivf_dbg_stop", stop)
                out["nlist%d_L%d_quad%d_stop%d" % (nlist, L, quad, stop)] = kernel_us(L)
            eng.set_option("ivf_dbg_stop", 0)
print(json.dumps(out))

# batch sweep: where four queries per block start to pay (one query per block leaves CUs idle below 1024 queries, and a CU's LDS to one block)
sweep = {}
cen = rng.integers(0, 256, size=(1024, M), dtype=np.uint8)
off, ids = bench.modulo_lists(N, 1024)
eng.set_posting_lists(cen, off, ids)
qbig = torch.from_numpy(rng.random((4096, M * 4)).astype(np.float32)).to(dev)
oi = torch.empty((4096, 1), dtype=torch.int64, device=dev); od = torch.empty((4096, 1), dtype=torch.float32, device=dev); oc = torch.empty((4096,), dtype=torch.int64, device=dev)
for Bs in (16, 64, 128, 256, 512, 1024, 2048, 4096):
    for quad in (1, 0):
        eng.set_option("ivf_quad", quad)
        def step(): eng.query_ivf_dev(qbig.data_ptr(), Bs, 1, 0, 0, 977, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), st.cuda_stream)
        for _ in range(50): step()
        torch.cuda.synchronize()
        eng.set_option("timing", 2); eng.timing_reset()
        for _ in range(100): step()
        torch.cuda.synchronize()
        ms, n = eng.timing_read("ivf_fused"); eng.set_option("timing", 0)
        sweep["B%d_quad%d" % (Bs, quad)] = round(ms / n * 1e3, 2)
eng.set_option("ivf_quad", 1)
print(json.dumps({"kernel_us_by_batch": sweep}))
