set -x
mkdir -p gpurun_out
rm -f gpurun_out/r3_wf_stagger.txt
for s in 0 1.5 3 6; do
  echo "== TMDIFF_WF_STAGGER $s (epi)" >> gpurun_out/r3_wf_stagger.txt
  TMDIFF_WF_STAGGER=$s timeout -k 10 120 python tools/bench_conv_wino.py 32 10 epi 2>&1 | grep -v amdgpu | cut -c1-60 >> gpurun_out/r3_wf_stagger.txt
done
cat gpurun_out/r3_wf_stagger.txt
