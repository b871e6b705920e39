python -m pytest tests/test_gpu_attention.py -m gpu -q -x > gpurun_out/r2_t17.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t17.log
python tools/bench_attention.py 2>&1 | grep attention
TMDIFF_ATTN_SIMPLE=1 python tools/bench_attention.py 2>&1 | grep attention
