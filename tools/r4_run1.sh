set -x
mkdir -p gpurun_out/r04
cd /root/repo
python -m pytest tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/t_round4.log
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/t_all.log
python tools/r4_measure.py gpurun_out/r04/measure1.json > gpurun_out/r04/measure1.log 2>&1
python bench.py > gpurun_out/r04/bench_default1.json 2> gpurun_out/r04/bench_default1.err
tail -5 gpurun_out/r04/t_round4.log; tail -5 gpurun_out/r04/t_all.log; tail -60 gpurun_out/r04/measure1.log
