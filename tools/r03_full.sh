#!/bin/bash
# full GPU suite, then the profile collection of tools/collect_profiles.sh under the given tag
set -o pipefail
tag=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1
rc=$?; tail -5 gpurun_out/${tag}_tests.log; [ $rc -ne 0 ] && exit $rc
bash tools/collect_profiles.sh $tag
