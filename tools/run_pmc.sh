#!/bin/bash
# run_pmc.sh OUT [fp32|bf16|"<script> <args...>"] : three rocprofv3 --pmc passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE) over
# tools/bench_conv.py 32 1 MATH (or over the given tools/ script), then tools/pmc_report.py -> gpurun_out/OUT.txt
R=$PWD; OUT=$1; WHAT=${2:-fp32}
case "$WHAT" in fp32|bf16) CMD="$R/tools/bench_conv.py 32 1 $WHAT";; *) CMD="$R/tools/$WHAT";; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY -d $R/gpurun_out/$OUT/main -o p --output-format csv -- python3 $CMD > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/$OUT/fetch -o p --output-format csv -- python3 $CMD > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/$OUT/write -o p --output-format csv -- python3 $CMD > /dev/null 2>&1
cd $R && python3 tools/pmc_report.py gpurun_out/$OUT/main gpurun_out/$OUT/fetch gpurun_out/$OUT/write > gpurun_out/$OUT.txt 2>&1
