#!/bin/bash
# bench.py over the workloads / shapes DESIGN.md section 8 quotes, one compact line each (run on the GPU box via gpurun)
p() { python -c "
import json,sys
for ln in sys.stdin:
    ln=ln.strip()
    if not ln.startswith('{'): continue
    l=json.loads(ln); r=l.get('roofline',{})
    ex={k:round(v,4) for k,v in r.items() if k.endswith('_ms_per_step')}
    print(l['config']['workload'][:88], '| qps', round(l['value']), '| ms', round(l['ms_per_step'],4), '| kern', r.get('kernel'), round(r.get('avg_launch_ms') or 0,4), 'x', r.get('launches_per_step'), 'frac', round(r.get('frac') or 0,3), ex, '| pipe', (l.get('pipelined') or {}).get('ms_per_step'), '| fresh', (l.get('fresh_queries') or {}).get('ms_per_step'), '| host', (l.get('host_call') or {}).get('ms_per_step'), '| p50/p99', l.get('p50_us'), l.get('p99_us'))
"; }
for a in "" "--topk 10" "--topk 100" "--workload subset" "--workload ivf" "--workload subset-ivf" "--workload ivf --topk 10" "--M 16" "--M 64" "--batch 4096" "--scan-mx 0" "--scan-mode 0" "$@"; do
  timeout 300 python bench.py $a --no-cpu-baseline 2>&1 | tail -1 | p
done
