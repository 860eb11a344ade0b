#!/bin/bash
# rocprofv3 kernel trace of the world-size-1 RCCL run (torchrun, one rank): what the sharded C-ABI step adds on the device -- the RCCL
# all-gather kernel, qshard_unpack_kernel / merge_top1_kernel -- beside the table / scan / re-rank kernels.   usage: tools/profile_world1.sh <tag> [bench args]
set -u
TAG=${1:-r04_world1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
OUT=gpurun_out/prof_summary; RAW=/tmp/rii_prof_w1
mkdir -p $OUT; rm -rf $RAW; mkdir -p $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt -o kt -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 \
    bench.py --gpus 1 --no-cpu-baseline --no-host-call --no-fresh --no-pipelined --no-live-counters --steps 20 --warmup 2 --preheat 0.05 "$@" > $OUT/${TAG}_bench_under_kernel_trace.json 2> $RAW/kt.err
python tools/summarize_prof.py stats $RAW/kt > $OUT/${TAG}_kernel_stats.txt
tail -3 $RAW/kt.err > $OUT/${TAG}_rocprof_stderr_tail.txt
rm -rf $RAW
head -30 $OUT/${TAG}_kernel_stats.txt
