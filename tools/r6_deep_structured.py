#!/usr/bin/env python3
"""Round 6: bench.py's `others.deep_structured` leg on its own (structured Deep1B-shaped set, real reconfigure, recall, reference)."""
import sys, os, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--no-cpu-baseline", action="store_true")
ap.add_argument("--levels", type=int, default=0)
ap.add_argument("--cap", type=int, default=0)
a = ap.parse_args()
args = argparse.Namespace(batch=a.batch, deep_structured=a.n, steps=a.steps, scan_mx=1, scan_mode=1, scan_order=1, no_cpu_baseline=a.no_cpu_baseline, table_levels=a.levels, cand_cap=a.cap)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
obj = bench.deep_structured_workload(args, torch, dev, "avx512", lambda: torch.cuda.synchronize())
print(json.dumps(obj))
