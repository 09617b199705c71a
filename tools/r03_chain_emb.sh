#!/bin/bash
# sampling with the embedding path on its own stream inside the interleaved chains (DM_CHAIN_EMB_SIDE=1) against the
# default (off), alternating on one box; then the generation tests with it on
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do for v in 0 1; do
  DM_CHAIN_EMB_SIDE=$v timeout -k 10 300 python bench.py --mode sample --sampler ddim --no-cpu-baseline > gpurun_out/chain_emb.json 2> gpurun_out/chain_emb.err || { tail -5 gpurun_out/chain_emb.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/chain_emb.json")); print("emb side in chains $v:", d["value"], d["unit"], d.get("ms_per_step"))
PY
done; done
DM_CHAIN_EMB_SIDE=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_model.py -q -x -k "generated or sampler or ddpm" --timeout 180 --timeout-method thread 2>&1 | tail -2
