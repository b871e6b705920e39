set -x
mkdir -p gpurun_out
TMDIFF_HIP_LIB=tools/lib_wstamps.so timeout -k 10 300 python tools/wino_stamps.py 32 > gpurun_out/r3_wstamps.txt 2>&1
TMDIFF_WINO_STAGGER=4 TMDIFF_HIP_LIB=tools/lib_wstamps.so timeout -k 10 300 python tools/wino_stamps.py 32 > gpurun_out/r3_wstamps_stagger4.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r3_t2.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3_t2.log
tail -5 gpurun_out/r3_t2.log
