#!/bin/bash
# Runs on the GPU box (through gpurun): the default benchmark line, a kernel trace of the same command, and the two
# HBM-traffic counter passes (each in a run of its own). Results land under gpurun_out/<tag>_*.
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o step -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sampling > $OUT/${TAG}_trace.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pmc_fetch -o step -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-sampling > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pmc_write -o step -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-sampling > $OUT/${TAG}_pmc_write.log 2>&1 || exit 1
echo collected
