set -x
mkdir -p gpurun_out
./tools/micro/cu_probe > gpurun_out/r3_cu_probe.txt 2>&1
for s in 0 3 6 10; do
  echo "== stagger $s" >> gpurun_out/r3_stagger.txt
  TMDIFF_WINO_STAGGER=$s timeout -k 10 120 python tools/bench_conv_wino.py 32 10 >> gpurun_out/r3_stagger.txt 2>&1
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t1.log 2>&1
echo "pytest rc $?" >> gpurun_out/r3_t1.log
tail -5 gpurun_out/r3_t1.log
