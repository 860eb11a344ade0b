p() { python -c "
import json,sys
for ln in sys.stdin:
    ln=ln.strip()
    if not ln.startswith('{'): continue
    l=json.loads(ln); r=l.get('roofline',{})
    print(l['config']['workload'][:70], '| qps', round(l['value']), '| ms', round(l['ms_per_step'],4), '| kern', r.get('kernel'), r.get('avg_launch_ms'), 'frac', r.get('frac'), '| pipe', (l.get('pipelined') or {}).get('ms_per_step'))
"; }
for a in "--topk 10" "--topk 100" "--workload subset" "--workload deep" "--M 16"; do timeout 300 python bench.py $a --no-cpu-baseline --no-host-call 2>&1 | tail -1 | p; done
