set -x
cd /root/repo
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r04/t9_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke9.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04/bench_default4.json 2> gpurun_out/r04/bench_default4.err
python bench.py --batch 128 --no-others --no-cpu-baseline --steps 50 > gpurun_out/r04/bench_b128_2.json 2> /dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 50 --warmup 5 --batch 128 --no-cpu-baseline > gpurun_out/r04/bench_world1_b128_2.json 2> /dev/null
tail -3 gpurun_out/r04/t9_all.log; tail -2 gpurun_out/r04/smoke9.log; tail -4 gpurun_out/r04/bench_default4.err
