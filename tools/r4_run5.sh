set -x
cd /root/repo
mkdir -p gpurun_out/r04
bash tools/profile_bench.sh r04_linear > gpurun_out/r04/prof_linear.log 2>&1
bash tools/profile_bench.sh r04_ivf --workload ivf > gpurun_out/r04/prof_ivf.log 2>&1
bash tools/profile_bench.sh r04_deep125m --workload deep --n-base 125000000 > gpurun_out/r04/prof_deep125m.log 2>&1
bash tools/profile_bench.sh r04_deep64m --workload deep --n-base 64000000 > gpurun_out/r04/prof_deep64m.log 2>&1
python bench.py --workload deep --n-base 125000000 --steps 10 --warmup 2 > gpurun_out/r04/bench_deep_125m.json 2> gpurun_out/r04/bench_deep_125m.err
ls gpurun_out/prof_summary | tail -30
tail -3 gpurun_out/r04/prof_deep125m.log
