set -x
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_capi.py tests/test_dist_gloo.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/t7.log
tail -8 gpurun_out/r04/t7.log
