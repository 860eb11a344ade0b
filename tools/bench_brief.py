#!/usr/bin/env python3
"""Runs bench.py with the given arguments (extra legs off) and prints a one-line digest: step, un-instrumented steps, dominant
kernel, per-kernel shares.   tools/bench_brief.py --topk 10"""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-others", "--no-host-call", "--no-fresh", "--no-pipelined"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True)
lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not lines:
    print("FAILED", out.stderr[-500:]); sys.exit(1)
d = json.loads(lines[-1]); r = d["roofline"]
print(" ".join(sys.argv[1:]) or "default", "| step %.4f ms | plain %s | %s %.4f ms frac %.3f | %s" % (
    d["ms_per_step"], [round(x, 4) for x in d["uninstrumented"]["ms_per_step"]], r["kernel"], r["avg_launch_ms"], r["frac"],
    {k: round(v, 4) for k, v in r.items() if k.endswith("_ms_per_step")}))
