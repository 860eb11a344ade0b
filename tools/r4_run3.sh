set -x
mkdir -p gpurun_out/r04
cd /root/repo
python -m pytest tests/test_gpu_round4.py -x -q -m gpu --durations=8 2>&1 | tail -30 > gpurun_out/r04/t3_round4.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "ivf or golden or small" 2>&1 | tail -8 > gpurun_out/r04/t3_ivf.log
python bench.py --workload ivf --steps 50 --no-cpu-baseline > gpurun_out/r04/bench_ivf1.json 2> gpurun_out/r04/bench_ivf1.err
python tools/readme_latency.py > gpurun_out/r04/readme_lat1.txt 2>&1
python tools/r4_deep_few.py 125000000 gpurun_out/r04/deep_few1.json > gpurun_out/r04/deep_few1.log 2>&1
tail -14 gpurun_out/r04/t3_round4.log; tail -4 gpurun_out/r04/t3_ivf.log; tail -12 gpurun_out/r04/readme_lat1.txt; tail -3 gpurun_out/r04/deep_few1.log
