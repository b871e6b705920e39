set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "winograd" > gpurun_out/r3_t3.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3_t3.log
tail -5 gpurun_out/r3_t3.log
grep -q "pytest rc 0" gpurun_out/r3_t3.log || exit 1
timeout -k 10 200 python tools/bench_conv_wino.py 32 10 > gpurun_out/r3_wf_bench.txt 2>&1
cat gpurun_out/r3_wf_bench.txt
TMDIFF_HIP_LIB=tools/lib_wfstamps.so timeout -k 10 300 python tools/wino_stamps.py 32 wf > gpurun_out/r3_wfstamps.txt 2>&1
