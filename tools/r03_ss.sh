#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do for n in 1 2 0; do
timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling --side-streams $n > gpurun_out/ss.json 2> gpurun_out/ss.err || { tail -5 gpurun_out/ss.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/ss.json")); print("side streams $n:", d["value"], d["ms_per_step"])
PY
done; done
