set -x
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r04/t8.log
python tools/opt_ab.py scan_pipe 0 1 > gpurun_out/r04/scan_pipe_ab.json 2> gpurun_out/r04/scan_pipe_ab.err
tail -4 gpurun_out/r04/t8.log; cat gpurun_out/r04/scan_pipe_ab.json
