#!/bin/bash
# copy_profiles.sh [TAG]: copy the summaries tools/collect_profiles.sh left under gpurun_out/ into profiles/ (tracked)
TAG=${1:-r04}; R=gpurun_out/${TAG}_final
cp gpurun_out/${TAG}_traffic.json profiles/${TAG}_bench_traffic.json
cp gpurun_out/${TAG}_traffic/stats/p_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
cp $R/prof_train/p_kernel_stats.csv profiles/${TAG}_train_step_kernel_stats.csv
cp $R/wgrad_b8.txt profiles/${TAG}_wgrad_b8.txt; cp $R/wgrad_b8_direct.txt profiles/${TAG}_wgrad_b8_direct.txt; cp $R/wgrad_b8_phases.txt profiles/${TAG}_wgrad_b8_phases.txt; cp $R/pmc_wgrad.txt profiles/${TAG}_wgrad_pmc.txt
cp $R/conv_layers_fp32.txt profiles/${TAG}_conv_layers_fp32.txt; cp $R/conv_layers_bf16.txt profiles/${TAG}_conv_layers_bf16.txt
cp $R/pmc_conv_fp32.txt profiles/${TAG}_conv_fp32_pmc.txt
cp $R/attention.txt profiles/${TAG}_attention.txt; cp $R/pmc_attention.txt profiles/${TAG}_attention_pmc.txt
cp $R/hbm_kernels.txt profiles/${TAG}_hbm_kernels.txt; grep config $R/configs.txt | grep -v Warning > profiles/${TAG}_configs.txt
tail -1 $R/bench_line.json > profiles/${TAG}_bench_line.json; tail -1 $R/bench_train_line.json > profiles/${TAG}_bench_train_line.json
grep "^{" $R/train_2rank_gloo.json > profiles/${TAG}_train_2rank_gloo.json
cp $R/conv_bf16_packed.txt profiles/${TAG}_conv_bf16_packed.txt; [ -f $R/conv_bf16_stamps.txt ] && cp $R/conv_bf16_stamps.txt profiles/${TAG}_conv_bf16_stamps.txt
cp $R/prof_bf16/p_kernel_stats.csv profiles/${TAG}_bf16_step_kernel_stats.csv
cp $R/conv_ll.txt profiles/${TAG}_conv_ll.txt; cp $R/pmc_conv_ll.txt profiles/${TAG}_conv_ll_pmc.txt
cp $R/conv_wino.txt profiles/${TAG}_conv_wino.txt; cp $R/pmc_conv_wino.txt profiles/${TAG}_conv_wino_pmc.txt
cp $R/conv_wino_epilogue.txt profiles/${TAG}_conv_wino_epilogue.txt; cp $R/bench_layers.txt profiles/${TAG}_bench_layers.txt
cp $R/cu_probe.txt profiles/${TAG}_cu_probe.txt; [ -f $R/conv_wf_ablation.txt ] && cp $R/conv_wf_ablation.txt profiles/${TAG}_conv_wf_ablation.txt
cp $R/wino_stamps.txt profiles/${TAG}_wino_stamps.txt; [ -f $R/wino_stamps_round2_kernel.txt ] && cp $R/wino_stamps_round2_kernel.txt profiles/${TAG}_wino_stamps_round2_kernel.txt
grep "^{" $R/bench_2rank_gloo.json > profiles/${TAG}_bench_2rank_gloo_line.json
cp $R/half_batches.txt profiles/${TAG}_half_batches.txt
