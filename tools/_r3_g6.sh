mkdir -p gpurun_out
for s in 0 3; do
  echo "== TMDIFF_WF_STAGGER $s" >> gpurun_out/r3_wfstamps2.txt
  TMDIFF_WF_STAGGER=$s TMDIFF_HIP_LIB=tools/lib_wfstamps.so timeout -k 10 300 python tools/wino_stamps.py 32 wf 2>&1 | grep -v amdgpu >> gpurun_out/r3_wfstamps2.txt
done
cat gpurun_out/r3_wfstamps2.txt | cut -c1-400
