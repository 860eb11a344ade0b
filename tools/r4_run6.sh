set -x
cd /root/repo
mkdir -p gpurun_out/r04
( time python bench.py ) > gpurun_out/r04/bench_default3.json 2> gpurun_out/r04/bench_default3.err
python bench.py --batch 128 --no-others --no-cpu-baseline --steps 50 > gpurun_out/r04/bench_b128.json 2> /dev/null
python tools/bench_brief.py --topk 10 > gpurun_out/r04/brief_top10.txt 2>&1
python tools/bench_brief.py --topk 100 > gpurun_out/r04/brief_top100.txt 2>&1
python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -25 > gpurun_out/r04/t6_all.log
tail -12 gpurun_out/r04/t6_all.log; tail -4 gpurun_out/r04/bench_default3.err; tail -3 gpurun_out/r04/brief_top100.txt
