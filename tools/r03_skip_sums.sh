#!/bin/bash
# upper bound of what hiding the weight-gradient slab sums could give: the step with those launches skipped
# (timing_skip=2; results are wrong, the timing is what is read) against the default, alternating on one box
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3; do for v in 0 2; do
  DM_HIP_CONFIG="timing_skip=$v" timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling > gpurun_out/skip.json 2> gpurun_out/skip.err || { tail -5 gpurun_out/skip.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/skip.json")); print("timing_skip=$v", d["value"], d["ms_per_step"])
PY
done; done
