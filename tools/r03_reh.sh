#!/bin/bash
# final-build check of the multi-GPU path on one box: the chain of graphs with a one-rank RCCL group issuing every
# collective, then the two-rank full-size rehearsal over gloo (both ranks on the one GPU)
set -o pipefail
mkdir -p gpurun_out
DM_DDP_DEBUG=1 timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling --force-collectives > gpurun_out/r03_force.json 2> gpurun_out/r03_force.err || { tail -5 gpurun_out/r03_force.err; exit 1; }
cut -c1-300 gpurun_out/r03_force.json; grep "\[segments\]" gpurun_out/r03_force.err | tail -1
DM_DDP_DEBUG=1 DM_DDP_TIMEOUT=60 DM_BENCH_WATCHDOG=200 DM_BENCH_BACKEND=gloo DM_BENCH_SHARE_GPU=1 timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 10 --warmup 3 --no-sampling > gpurun_out/r03_reh.json 2> gpurun_out/r03_reh.err
rc=$?; echo "rehearsal rc=$rc"; cut -c1-400 gpurun_out/r03_reh.json; tail -5 gpurun_out/r03_reh.err
exit $rc
