#!/bin/bash
# bench_traffic.sh [TAG]: HBM traffic of the benchmark's 3x3x3 conv launches from PMC counters, collected as
# MI355X_MICROARCH.md prescribes: separate `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE do not fit one pass),
# only --kernel-trace beside them, the program itself after `--`.  Writes profiles/${TAG}_bench_traffic.json (read back
# by bench.py into roofline.traffic) and a kernel-stats pass of the same command (profiles/${TAG}_bench_kernel_stats.csv).
R=$PWD; TAG=${1:-r04}; OUT=$R/gpurun_out/${TAG}_traffic
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-extras"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- $CMD > $OUT.fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- $CMD > $OUT.write.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- $CMD > $OUT.stats.log 2>&1
rc=$?
cd $R && python3 tools/bench_traffic.py $OUT $TAG && cp $(ls $OUT/stats/*/*kernel_stats.csv $OUT/stats/*kernel_stats.csv 2>/dev/null | head -1) profiles/${TAG}_bench_kernel_stats.csv
# (on a gpurun box only gpurun_out/ travels back: ${OUT}.json and $OUT/stats/p_kernel_stats.csv are the copies to commit under profiles/)
exit $rc
