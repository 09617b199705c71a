#!/bin/bash
# every shipped configuration through the benchmark on the final build (short runs)
set -o pipefail
mkdir -p gpurun_out
for c in beta_cifar rf_cifar rf_cifar_long rf_mnist fid_test; do
  timeout -k 10 300 python bench.py --config $c --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/cfg_$c.json 2> gpurun_out/cfg_$c.err || { tail -5 gpurun_out/cfg_$c.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/cfg_$c.json"))
print("$c", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["wgrad"]["frac"], [(s["sampler"], s["n_steps"], s["img_s"]) for s in d["sampling"]])
PY
done
