#!/usr/bin/env python3
"""Round 5: the inverted index through a one-rank RCCL communicator -- plain rii_query_ivf_dev against the query-sharded and the
database-sharded entry points, B = 1024 and 128, SIFT shape (M = 32, nlist = 1024, L = 977; random codes, modulo partition) and
per-kernel shares of the database-sharded step (timing = 1)."""
import sys, json, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rii_amd import RiiGpu
from rii_amd import dist as rd
dev = torch.device("cuda", 0)
N, M, nlist, L = 1_000_000, 32, 1024, 977
rng = np.random.default_rng(1)
cw = rng.random((M, 256, 4)).astype(np.float32)
codes = rng.integers(0, 256, size=(N, M), dtype=np.uint8)
eng = RiiGpu(cw, False, device=0); eng.add_codes(codes, False)
if len(sys.argv) > 1:                                    # phase split: the shard kernel cut short (wrong rows)
    eng.set_option("shard_dbg_stop", int(sys.argv[1]))
off, ids = bench.modulo_lists(N, nlist)
eng.set_posting_lists(rng.integers(0, 256, size=(nlist, M), dtype=np.uint8), off, ids)
comm = rd.get_comm()
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
out = {}
for B in (1024, 128):
    q = torch.from_numpy(rng.random((B, M * 4)).astype(np.float32)).to(dev)
    mk = lambda: (torch.empty((B, 1), dtype=torch.int64, device=dev), torch.empty((B, 1), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.int64, device=dev))
    ri, rdd, rc = mk(); oi, od, oc = mk()
    fns = {"plain": lambda: eng.query_ivf_dev(q.data_ptr(), B, 1, 0, 0, L, ri.data_ptr(), rdd.data_ptr(), rc.data_ptr(), s),
           "query_sharded": lambda: comm.query_ivf_qsharded_dev(eng, q.data_ptr(), B, 1, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), s),
           "db_sharded": lambda: comm.query_ivf_dbsharded_dev(eng, 0, N, q.data_ptr(), B, 1, 0, 0, 0, L, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), 0, s)}
    res = {}
    for name, fn in fns.items():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1: fn(); torch.cuda.synchronize()
        K = 200
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K): fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / K * 1e6)
        res[name] = round(best, 2)
        if name != "plain": res[name + "_match"] = bool(torch.equal(oi, ri) and torch.equal(od, rdd))
    eng.set_option("timing", 1); eng.timing_reset()
    for _ in range(50): fns["db_sharded"]()
    torch.cuda.synchronize()
    res["db_sharded_kernel_us"] = {k: round(eng.timing_read(k)[0] / max(eng.timing_read(k)[1], 1) * 1e3, 2) for k in ("ivf_shard", "lut") if eng.timing_read(k)[1]}
    eng.set_option("timing", 0)
    out["B%d" % B] = res
print(json.dumps(out))
