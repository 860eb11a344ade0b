set -x
mkdir -p gpurun_out/r04
cd /root/repo
python -m pytest tests/test_gpu_round4.py -x -q -m gpu --durations=5 2>&1 | tail -20 > gpurun_out/r04/t4_round4.log
python tools/r4_ivf_ab.py ivf_inline_exact 0 1 gpurun_out/r04/ivf_inline_ab.json > gpurun_out/r04/ivf_inline_ab.log 2>&1
python bench.py --latency --batch 1 --topk 3 --no-cpu-baseline > gpurun_out/r04/latency_n1m_top3.json 2> gpurun_out/r04/latency_n1m_top3.err
python bench.py --latency --batch 1 --topk 1 --no-cpu-baseline > gpurun_out/r04/latency_n1m_top1.json 2> /dev/null
python bench.py --latency --batch 4 --topk 10 --no-cpu-baseline > gpurun_out/r04/latency_n1m_b4_top10.json 2> /dev/null
tail -12 gpurun_out/r04/t4_round4.log; tail -2 gpurun_out/r04/ivf_inline_ab.log; cat gpurun_out/r04/latency_n1m_top3.json gpurun_out/r04/latency_n1m_top1.json gpurun_out/r04/latency_n1m_b4_top10.json | grep -o '"latency": {.*}}' 
