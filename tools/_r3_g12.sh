mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "winograd" 2>&1 | tail -3
timeout -k 10 120 python tools/bench_conv_wino.py 32 10 2>&1 | grep -v amdgpu | cut -c1-48
timeout -k 10 120 python tools/bench_conv_wino.py 32 10 epi 2>&1 | grep -v amdgpu | cut -c1-48
bash tools/run_pmc.sh r3_pmc_wf2 "bench_conv_wino.py 32 1"
grep "wf<" gpurun_out/r3_pmc_wf2.txt | cut -c1-220
