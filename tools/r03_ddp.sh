#!/bin/bash
# one box: single graph, graph chain cut at the bucket boundaries, and the chain with a one-rank RCCL group issuing every collective
set -o pipefail
mkdir -p gpurun_out
for v in single split force; do
  flag=""; [ $v = split ] && flag="--split-graph"; [ $v = force ] && flag="--force-collectives"
  DM_DDP_DEBUG=1 timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling $flag > gpurun_out/r03_ddp_$v.json 2> gpurun_out/r03_ddp_$v.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r03_ddp_$v.json")); print("$v", d["value"], d["ms_per_step"], d["config"].get("graph_segments"))
PY
  grep "\[segments\]" gpurun_out/r03_ddp_$v.err | tail -1
done
