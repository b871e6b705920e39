mkdir -p gpurun_out
echo "== normal" > gpurun_out/r3_wf_ablate.txt
timeout -k 10 120 python tools/bench_conv_wino.py 32 10 2>&1 | grep -v amdgpu | cut -c1-48 >> gpurun_out/r3_wf_ablate.txt
for v in 1 2 3; do
echo "== ablate $v" >> gpurun_out/r3_wf_ablate.txt
TMDIFF_HIP_LIB=tools/lib_wfab$v.so timeout -k 10 120 python tools/bench_conv_wino.py 32 10 2>&1 | grep -v amdgpu | cut -c1-48 >> gpurun_out/r3_wf_ablate.txt
done
cat gpurun_out/r3_wf_ablate.txt
