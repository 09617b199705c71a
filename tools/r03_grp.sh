#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for g in 0 2 4 0; do
DM_WGRAD_GROUP=$g timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling > gpurun_out/grp.json 2> gpurun_out/grp.err || { tail -5 gpurun_out/grp.err; exit 1; }
python - <<PY
import json
d = json.load(open("gpurun_out/grp.json")); print("group $g:", d["value"], d["ms_per_step"], d["roofline"]["wgrad"]["ms_per_step"])
PY
done
