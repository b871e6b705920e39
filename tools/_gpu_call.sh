python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py tests/test_gpu_kernels.py -m gpu -q -x > gpurun_out/r2_t8.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t8.log
python tools/bench_wgrad.py 8 3 > gpurun_out/r2_wgrad_b8_v3.txt 2>&1; tail -15 gpurun_out/r2_wgrad_b8_v3.txt
python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | cut -c1-400
TMDIFF_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --mode train --steps 4 --warmup 2 > gpurun_out/r2_train_2rank_gloo.json 2> gpurun_out/r2_train_2rank_gloo.err; echo "2-rank gloo train rc=$?"; cat gpurun_out/r2_train_2rank_gloo.json; tail -5 gpurun_out/r2_train_2rank_gloo.err
