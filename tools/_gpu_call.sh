python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r2_t9.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t9.log
python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | cut -c1-900
