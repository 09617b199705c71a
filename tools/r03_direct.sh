#!/bin/bash
# Patch kernel's register epilogue against its LDS epilogue: tests, per-layer times from cold caches, the step A/B.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_ops.py -x -q -m gpu --timeout 120 --timeout-method thread -k "conv" > gpurun_out/direct_tests.log 2>&1 || { tail -30 gpurun_out/direct_tests.log; exit 1; }
tail -3 gpurun_out/direct_tests.log
for v in 0 1; do
  echo "== patch_direct=$v (cold)"
  DM_BENCH_COLD=1 DM_HIP_CONFIG="patch_direct=$v" timeout -k 10 300 python tools/bench_conv.py "" fd 2>&1 | grep -v "1x1\|s2\|attn" || exit 1
done
bash tools/ab.sh direct "patch_direct=0" "patch_direct=1" --no-sampling
