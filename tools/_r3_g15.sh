mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "winograd" 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_e.json 2> gpurun_out/r3_bench_e.err
timeout -k 10 300 python tools/bench_configs.py c5 2>&1 | grep -i config
