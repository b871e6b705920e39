python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k config4 -s 2>&1 | grep -v Warning | tail -25
