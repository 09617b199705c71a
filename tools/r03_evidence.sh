#!/bin/bash
# The round's evidence on one box: default benchmark line, kernel trace, HBM-traffic and matrix-pipe counter passes,
# reduced to their summaries on the box (the result databases are too large to bring back).
set -u
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
bash $R/tools/collect_profiles.sh $TAG || exit 1
finddb() { ls $OUT/$1/*/*.db $OUT/$1/*.db 2>/dev/null | head -1; }
python3 $R/tools/prof_summary.py $(finddb ${TAG}_trace) 30 > $OUT/${TAG}_step_kernel_stats_table.md || exit 1
python3 $R/tools/traffic_summary.py $(finddb ${TAG}_pmc_fetch) $(finddb ${TAG}_pmc_write) $OUT/${TAG}_conv_traffic.json > $OUT/${TAG}_conv_traffic_table.md || exit 1
bash $R/tools/r03_pmc.sh $TAG > /dev/null || exit 1
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma
echo evidence collected
