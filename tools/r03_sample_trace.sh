#!/bin/bash
# kernel trace of the sampling loop (DDIM-50, B=256): where a UNet forward's time goes
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/r03_sample_trace -o s -- python3 $R/bench.py --mode sample --no-cpu-baseline > $OUT/r03_sample_trace.log 2>&1 || { tail -5 $OUT/r03_sample_trace.log; exit 1; }
db=$(ls $OUT/r03_sample_trace/*/*.db $OUT/r03_sample_trace/*.db 2>/dev/null | head -1)
python3 $R/tools/prof_summary.py $db 1 > $OUT/r03_sample_kernel_stats_table.md || exit 1
rm -rf $OUT/r03_sample_trace
head -40 $OUT/r03_sample_kernel_stats_table.md; tail -2 $OUT/r03_sample_kernel_stats_table.md; tail -3 $OUT/r03_sample_trace.log | cut -c1-600
