#!/bin/bash
# back-to-back graph launches: rate of differing pairs, and the first differing tensor, per build of the library
# usage: tools/r03_diag.sh <lib:hooks:pairs> ...   lib = tree | old | fast | <name> for tools/bin/libdm_hip_<name>.so
set -o pipefail
mkdir -p gpurun_out
for spec in "$@"; do
  IFS=: read -r lib hooks pairs <<< "$spec"
  if [ "$lib" = tree ]; then unset DM_HIP_LIBRARY; else export DM_HIP_LIBRARY=$PWD/tools/bin/libdm_hip_$lib.so; fi
  echo "== $spec"
  DIAG_HOOKS=$hooks timeout -k 10 400 python -u tools/diag_localise.py $pairs plain 2>&1 | grep -v "Warning\|amdgpu.ids\|float(la)" | tee gpurun_out/diag_${lib}_$hooks.log | cut -c1-150 | tail -25 || exit 1
done
