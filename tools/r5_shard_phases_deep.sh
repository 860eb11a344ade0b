# phase split of ivf_shard_any_kernel at the Deep1B-shaped shard (64 M codes, nlist = L = 8000): kernel time with the kernel cut short
cd /root/repo
for s in 1 2 3 4 5 0; do echo -n "stop $s: "; timeout 250 python bench.py --workload deep-ivf --shard-dbg-stop $s --n-base ${1:-64000000} --steps 5 --warmup 1 --no-cpu-baseline ${2:+--batch $2} 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4))"; done
