#!/bin/bash
# What each kernel family costs the step end to end, on one box: the step with the family's launches issued TWICE
# (timing_twice: data stay valid), and with them skipped (timing_skip: garbage flows on, the chip clocks differently).
set -o pipefail
mkdir -p gpurun_out
run() {
  DM_HIP_CONFIG="$1" timeout -k 10 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-sampling > gpurun_out/abl.json 2> gpurun_out/abl.err || { tail -3 gpurun_out/abl.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abl.json")); print("$1", d["ms_per_step"])
PY
}
run timing_twice=0
for m in 1 2 4 8 16 32 64 128; do run timing_twice=$m; done
run timing_twice=0
for m in 8 16 24; do run timing_skip=$m; done
