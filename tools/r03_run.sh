#!/bin/bash
# usage: tools/r03_run.sh <tag> <pytest selection...>   - GPU tests, then one bench line with per-launch dump
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest "$@" -x -q --timeout 240 --timeout-method thread > gpurun_out/${tag}_t.log 2>&1
rc=$?; tail -4 gpurun_out/${tag}_t.log; [ $rc -ne 0 ] && exit $rc
DM_BENCH_DUMP_CONVS=1 timeout -k 10 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_b.json 2> gpurun_out/${tag}_b.err
rc=$?; python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_b.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_step"], d["roofline"]["wgrad"]["frac"], d["roofline"]["wgrad"]["ms_per_step"])
for r in d["sampling"]: print(r["sampler"], r["n_steps"], r["img_s"], r["unet_forward_ms"])
PY
exit $rc
