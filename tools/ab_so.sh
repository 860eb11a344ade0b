#!/bin/bash
# Same-box A/B of two builds of librii_amd.so (tools/_ab_old.so, tools/_ab_new.so: built here, travel with gpurun, git-ignored):
# alternates them under tools/bench_brief.py with the given arguments.   usage: tools/ab_so.sh [rounds] [bench_brief args...]
R=${1:-2}; shift || true
cp rii_amd/librii_amd.so /tmp/keep.so
for i in $(seq $R); do
  for v in old new; do
    cp tools/_ab_$v.so rii_amd/librii_amd.so
    echo -n "$v: "; timeout -s KILL 100 python tools/bench_brief.py "$@" < /dev/null
  done
done
cp /tmp/keep.so rii_amd/librii_amd.so
