set -x
mkdir -p gpurun_out
timeout -k 10 200 python tools/bench_conv_wino.py 32 10 > gpurun_out/r3_wf_bench2.txt 2>&1
cat gpurun_out/r3_wf_bench2.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_wf.json 2> gpurun_out/r3_bench_wf.err
TMDIFF_WF=0 timeout -k 10 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_nowf.json 2> gpurun_out/r3_bench_nowf.err
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r3_t4.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3_t4.log
tail -5 gpurun_out/r3_t4.log
