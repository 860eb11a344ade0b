#!/bin/bash
# profiles/r04_latency_n1m.json: one-query (and four-query) calls at N = 1M through bench.py --latency.   usage: tools/r4_latency_n1m.sh out.json
out=${1:-gpurun_out/r04/latency_n1m.json}
python - "$out" <<'PY'
import json, subprocess, sys
res = {}
for name, args in (("top3", ["--batch", "1", "--topk", "3"]), ("top1", ["--batch", "1", "--topk", "1"]), ("b4_top10", ["--batch", "4", "--topk", "10"])):
    p = subprocess.run([sys.executable, "bench.py", "--latency", "--steps", "300", "--no-cpu-baseline"] + args, capture_output=True, text=True)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    res[name] = {"config": d["config"]["workload"] if isinstance(d.get("config"), dict) else str(d.get("config")), "latency": d.get("latency")}
json.dump(res, open(sys.argv[1], "w"), indent=0)
print(json.dumps({k: {kk: round(vv["p50_ms"] * 1e3, 1) for kk, vv in v["latency"].items()} for k, v in res.items()}))
PY
