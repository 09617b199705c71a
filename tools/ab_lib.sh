#!/bin/bash
# the library in the tree against another build of it (tools/bin/libdm_hip_old.so), alternating, on one box
set -o pipefail
mkdir -p gpurun_out
OLD=$PWD/tools/bin/libdm_hip_old.so
for i in 1 2 3; do for v in old new; do
  lib=""; [ $v = old ] && lib=$OLD
  DM_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling "$@" > gpurun_out/ablib.json 2> gpurun_out/ablib.err || { tail -5 gpurun_out/ablib.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/ablib.json")); print("$v", d["value"], d["ms_per_step"])
PY
done; done
