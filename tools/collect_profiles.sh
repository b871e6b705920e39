#!/bin/bash
# collect_profiles.sh [TAG]: everything profiles/ holds for a round, in one GPU call (~4 min).  Outputs land in
# gpurun_out/${TAG}_final/ (scratch); copy the summaries into profiles/ afterwards (tools/README.md).
R=$PWD; TAG=${1:-r04}; O=$R/gpurun_out/${TAG}_final; mkdir -p $O
# (the PMC traffic first: bench.py reads profiles/${TAG}_bench_traffic.json back into roofline.traffic and marks it stale when it
#  was collected from other sources)
bash tools/bench_traffic.sh $TAG > $O/traffic.log 2>&1; echo "traffic rc=$?"
python3 bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python3 bench.py --mode train --steps 10 --warmup 3 > $O/bench_train_line.json 2>> $O/bench.err; echo "train rc=$?"
python3 tools/bench_configs.py c1 c3 c4 2>&1 | grep config > $O/configs.txt
python3 tools/bench_attention.py 2>&1 | grep -v amdgpu > $O/attention.txt
python3 tools/bench_hbm_kernels.py 2>&1 | grep -v amdgpu > $O/hbm_kernels.txt
python3 tools/bench_wgrad.py 8 5 plain 2>&1 | grep -v amdgpu > $O/wgrad_b8.txt
TMDIFF_WGRAD_WINO=0 python3 tools/bench_wgrad.py 8 5 plain 2>&1 | grep -v amdgpu > $O/wgrad_b8_direct.txt
for ph in 1 2 4; do echo "== TMDIFF_WW_PHASES=$ph (1: the two transform passes, 2: the accumulation kernel, 4: the reduction; alone, results stale)"; TMDIFF_WW_PHASES=$ph python3 tools/bench_wgrad.py 8 20 plain 2>&1 | grep -v amdgpu | cut -c1-48; done > $O/wgrad_b8_phases.txt
python3 tools/bench_conv.py 32 3 fp32 2>&1 | grep -v amdgpu > $O/conv_layers_fp32.txt
python3 tools/bench_conv.py 32 3 bf16 2>&1 | grep -v amdgpu > $O/conv_layers_bf16.txt
python3 tools/bench_conv_ll.py 32 5 2>&1 | grep -v amdgpu > $O/conv_ll.txt
python3 tools/bench_conv_wino.py 32 5 2>&1 | grep -v amdgpu > $O/conv_wino.txt
python3 tools/bench_conv_wino.py 32 5 epi 2>&1 | grep -v amdgpu > $O/conv_wino_epilogue.txt
python3 tools/bench_layers.py 5 2>&1 | grep -v amdgpu > $O/bench_layers.txt
# (built here from its source: no binary is kept in the tree)
hipcc -O2 --offload-arch=gfx950 -o tools/micro/cu_probe tools/micro/cu_probe.hip && ./tools/micro/cu_probe > $O/cu_probe.txt 2>&1
# (diagnostic builds with s_memrealtime stamps / ablation switches, before the call:
#  tools/build_variant.sh wstamps "-DTMDIFF_WINO_STAMPS=1" conv3d_wino; tools/build_variant.sh wfstamps "-DTMDIFF_WF_STAMPS=1" conv3d_wf;
#  for v in 1 2 3: tools/build_variant.sh wfab$v "-DTMDIFF_WF_ABLATE=$v" conv3d_wf)
[ -f tools/lib_wstamps.so ] && TMDIFF_HIP_LIB=tools/lib_wstamps.so python3 tools/wino_stamps.py 32 2>&1 | grep -v amdgpu > $O/wino_stamps_round2_kernel.txt
[ -f tools/lib_wfstamps.so ] && TMDIFF_HIP_LIB=tools/lib_wfstamps.so python3 tools/wino_stamps.py 32 wf 2>&1 | grep -v amdgpu > $O/wino_stamps.txt
for v in 1 2 3; do [ -f tools/lib_wfab$v.so ] && { echo "== TMDIFF_WF_ABLATE=$v (1: no input transform, 2: + no raw DMA, 3: + no weight DMA; results wrong)"; TMDIFF_HIP_LIB=tools/lib_wfab$v.so python3 tools/bench_conv_wino.py 32 5 2>&1 | grep -v amdgpu | cut -c1-48; } >> $O/conv_wf_ablation.txt; done
python3 tools/bench_bf16_dma.py 32 10 2>&1 | grep -v amdgpu > $O/conv_bf16_packed.txt
# (diagnostic build with s_memtime stamps: tools/build_variant.sh stamps "-DTMDIFF_BF16_STAMPS=1" conv3d_bf16, before the call)
[ -f tools/lib_stamps.so ] && TMDIFF_HIP_LIB=tools/lib_stamps.so python3 tools/bf16_stamps.py 32 2>&1 | grep -v amdgpu > $O/conv_bf16_stamps.txt
bash tools/run_pmc.sh ${TAG}_final/pmc_wgrad "bench_wgrad.py 8 1 plain"
bash tools/run_pmc.sh ${TAG}_final/pmc_attention "bench_attention.py"
bash tools/run_pmc.sh ${TAG}_final/pmc_conv_fp32 fp32
bash tools/run_pmc.sh ${TAG}_final/pmc_conv_ll "bench_conv_ll.py 32 1"
bash tools/run_pmc.sh ${TAG}_final/pmc_conv_wino "bench_conv_wino.py 32 1"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_train -o p --output-format csv -- python3 $R/bench.py --mode train --steps 6 --warmup 2 > /dev/null 2>&1; echo "train prof rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_bf16 -o p --output-format csv -- python3 $R/tools/prof_bf16_step.py 10 > /dev/null 2>&1; echo "bf16 prof rc=$?"
TMDIFF_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --mode train --steps 4 --warmup 2 > $O/train_2rank_gloo.json 2> /dev/null; echo "2-rank rc=$?"
TMDIFF_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --steps 5 --warmup 2 --no-extras > $O/bench_2rank_gloo.json 2> /dev/null; echo "2-rank sample rc=$?"
python3 $R/tools/exp_half_batches.py 10 2>&1 | grep -v amdgpu > $O/half_batches.txt
