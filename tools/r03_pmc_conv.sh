#!/bin/bash
# wave-state counters of the patch kernel alone: 64-wide tiles (8x8 512->512) against 128-wide ones (16x16 512->512)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for s in "L8 512->512" "L16 up 512->512" "L32 128->128"; do
  tag=$(echo "$s" | tr -c 'A-Za-z0-9' '_')
  timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmcc_$tag -o r -- python3 $R/tools/bench_conv.py "$s" dw > $OUT/pmcc_$tag.log 2>&1 || exit 1
  echo "== $s"
  python3 $R/tools/pmc_summary.py $(ls $OUT/pmcc_$tag/*.db $OUT/pmcc_$tag/*/*.db 2>/dev/null | head -1) conv_ | sed 's/(anonymous namespace):://' | cut -c1-140
done
