python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py tests/test_gpu_backward.py -m gpu -q -x > gpurun_out/r2_t11.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t11.log
python tools/bench_wgrad.py 8 3 2>&1 | tail -4
python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | cut -c1-120
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2_prof_train3 -o p --output-format csv -- python3 $R/bench.py --mode train --steps 6 --warmup 2 > /dev/null 2>&1; echo "prof rc=$?"
