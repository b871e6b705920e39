mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r3_bench_c.json 2> gpurun_out/r3_bench_c.err
echo "bench rc $?"
bash tools/run_pmc.sh r3_pmc_wf "bench_conv_wino.py 32 1"
cat gpurun_out/r3_pmc_wf.txt | cut -c1-260
