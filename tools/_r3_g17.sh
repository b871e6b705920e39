mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python tools/bench_wgrad.py 8 3 2>&1 | grep -v amdgpu
