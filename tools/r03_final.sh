#!/bin/bash
# full GPU suite, smoke(), profile collection and the matrix-pipe counters, in one call
set -o pipefail
tag=${1:-r03}
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${tag}_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
bash tools/collect_profiles.sh $tag || exit 1
bash tools/r03_pmc.sh $tag || exit 1
