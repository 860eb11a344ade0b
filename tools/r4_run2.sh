set -x
mkdir -p gpurun_out/r04
cd /root/repo
python -m pytest tests/test_gpu_round4.py tests/test_capi.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04/t2_round4.log
python -m pytest tests/test_dist_gloo.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04/t2_dist.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench_world1.json 2> gpurun_out/r04/bench_world1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 50 --warmup 5 --batch 128 --no-cpu-baseline > gpurun_out/r04/bench_world1_b128.json 2> gpurun_out/r04/bench_world1_b128.err
( time python bench.py ) > gpurun_out/r04/bench_default2.json 2> gpurun_out/r04/bench_default2.err
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/t2_all.log
tail -8 gpurun_out/r04/t2_round4.log; tail -8 gpurun_out/r04/t2_dist.log; tail -5 gpurun_out/r04/t2_all.log; tail -3 gpurun_out/r04/bench_world1.err; tail -5 gpurun_out/r04/bench_default2.err
