#!/bin/bash
# run_pmc.sh OUT MATH : three rocprofv3 --pmc passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE) over tools/bench_conv.py 32 1 MATH
R=$PWD; OUT=$1; MATH=${2:-fp32}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/$OUT/main -o p --output-format csv -- python3 $R/tools/bench_conv.py 32 1 $MATH > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/$OUT/fetch -o p --output-format csv -- python3 $R/tools/bench_conv.py 32 1 $MATH > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/$OUT/write -o p --output-format csv -- python3 $R/tools/bench_conv.py 32 1 $MATH > /dev/null 2>&1
cd $R && python3 tools/pmc_report.py gpurun_out/$OUT/main gpurun_out/$OUT/fetch gpurun_out/$OUT/write > gpurun_out/$OUT.txt 2>&1
