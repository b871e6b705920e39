mkdir -p gpurun_out
timeout -k 10 120 python tools/bench_conv_wino.py 32 10 2>&1 | grep -v amdgpu | cut -c1-48
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_t10.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3_t10.log
tail -8 gpurun_out/r3_t10.log
