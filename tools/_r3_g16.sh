mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_layers.py 5 c3 bf16 2>&1 | grep -v amdgpu > gpurun_out/r3_layers_c3_bf16.txt
head -40 gpurun_out/r3_layers_c3_bf16.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r3_c3bf16_prof -o p --output-format csv -- python3 tools/bench_layers.py 3 c3 bf16 > /dev/null 2>&1
head -25 gpurun_out/r3_c3bf16_prof/p_kernel_stats.csv | cut -c1-150
