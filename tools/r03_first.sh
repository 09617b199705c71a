#!/bin/bash
# DDP + harness tests, one bench line, then the two-rank full-size rehearsal over gloo
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_ddp_gpu.py tests/test_gpu_harness.py tests/test_zz_no_alloc.py "tests/test_gpu_ops.py::test_adamw_ema_step_matches_torch" "tests/test_gpu_model.py" -x -q --timeout 180 --timeout-method thread > gpurun_out/r03_t1.log 2>&1
rc=$?; tail -5 gpurun_out/r03_t1.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 30 --warmup 10 > gpurun_out/r03_b1.json 2> gpurun_out/r03_b1.err
rc=$?; cut -c1-600 gpurun_out/r03_b1.json; [ $rc -ne 0 ] && exit $rc
DM_DDP_DEBUG=1 DM_DDP_TIMEOUT=60 DM_BENCH_WATCHDOG=200 DM_BENCH_BACKEND=gloo DM_BENCH_SHARE_GPU=1 timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 10 --warmup 3 --no-sampling > gpurun_out/r03_reh.json 2> gpurun_out/r03_reh.err
rc=$?; echo "rehearsal rc=$rc"; cut -c1-400 gpurun_out/r03_reh.json; tail -5 gpurun_out/r03_reh.err
exit 0
