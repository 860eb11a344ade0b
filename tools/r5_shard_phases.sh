cd /root/repo
for s in 1 2 3 4 5 0; do echo -n "stop $s: "; timeout 200 python tools/r5_sharded_ivf.py $s 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print({b:(d[b]['db_sharded'], d[b]['db_sharded_kernel_us']) for b in d})"; done
