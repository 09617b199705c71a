#!/bin/bash
# one-box sweep of run-time switches: step time and the weight-gradient / convolution legs of the roofline object
set -o pipefail
mkdir -p gpurun_out
for cfg in "$@"; do
  DM_HIP_CONFIG="$cfg" timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/sweep.json"))
r = d["roofline"]
print("$cfg", d["ms_per_step"], "conv", r["ms_per_step"], r["frac"], "wgrad", r["wgrad"]["ms_per_step"], r["wgrad"]["frac"])
PY
done
