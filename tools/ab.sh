#!/bin/bash
# usage: tools/ab.sh <tag> "<DM_HIP_CONFIG of A>" "<DM_HIP_CONFIG of B>" [extra bench args]: A/B/A/B on one box
set -o pipefail
tag=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
for i in 1 2; do
  for v in A B; do
    cfg=$A; [ $v = B ] && cfg=$B
    DM_HIP_CONFIG="$cfg" timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline "$@" > gpurun_out/${tag}_${v}${i}.json 2> gpurun_out/${tag}_${v}${i}.err || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_${v}${i}.json"))
s=d.get("sampling") or []
print("$v$i [$cfg]", d["value"], d["ms_per_step"], d["roofline"]["ms_per_step"] if d.get("roofline") else None, [ (r["n_steps"], r["img_s"], r["unet_forward_ms"]) for r in s])
PY
  done
done
