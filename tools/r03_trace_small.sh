#!/bin/bash
# short kernel trace of the training step, summary only (rows matching $1)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/tr_small -o s -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sampling > $OUT/tr_small.log 2>&1 || { tail -5 $OUT/tr_small.log; exit 1; }
db=$(ls $OUT/tr_small/*/*.db $OUT/tr_small/*.db 2>/dev/null | head -1)
python3 $R/tools/prof_summary.py $db 30 > $OUT/tr_small_table.md || exit 1
rm -rf $OUT/tr_small
grep -E "${1:-.}" $OUT/tr_small_table.md; tail -1 $OUT/tr_small_table.md
