#!/bin/bash
# matrix-pipe counters, then the convolutions per launch inside a step (HIP events) and alone (warm and cold caches)
set -o pipefail
mkdir -p gpurun_out
bash tools/r03_pmc.sh r03 || exit 1
cd $GRAFT_REPO_ROOT
DM_BENCH_DUMP_CONVS=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sampling > gpurun_out/r03_dump.json 2> gpurun_out/r03_dump.err || exit 1
timeout -k 10 300 python tools/bench_conv.py > gpurun_out/r03_conv_warm.txt 2>&1 || exit 1
DM_BENCH_COLD=1 timeout -k 10 300 python tools/bench_conv.py > gpurun_out/r03_conv_cold.txt 2>&1 || exit 1
echo done
