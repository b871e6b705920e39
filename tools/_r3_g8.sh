mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_layers.py 5 2>&1 | grep -v amdgpu > gpurun_out/r3_layers.txt
cat gpurun_out/r3_layers.txt
