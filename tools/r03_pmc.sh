#!/bin/bash
# matrix-pipe busy counters of every kernel of the training step (one --pmc pass, program directly after --)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_mfma -o step -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-sampling > $OUT/${TAG}_pmc_mfma.log 2>&1 || exit 1
db=$(ls $OUT/${TAG}_pmc_mfma/*/*.db $OUT/${TAG}_pmc_mfma/*.db 2>/dev/null | head -1)
python3 $R/tools/mfma_busy_md.py $db > $OUT/${TAG}_mfma_busy_table.md || exit 1
cat $OUT/${TAG}_mfma_busy_table.md
