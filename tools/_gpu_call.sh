python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py -m gpu -q -x > gpurun_out/r2_t10.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t10.log
python tools/bench_conv.py 32 3 fp32 2>&1 | grep "k1"
TMDIFF_CONV1_DWORD=1 python tools/bench_conv.py 32 3 fp32 2>&1 | grep "k1"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['k1_conv'], d['bf16_compute']['value'], d['train_step']['ms_per_step'])"
