#!/bin/bash
set -o pipefail
tag=${1:-r03}
bash tools/collect_profiles.sh $tag || exit 1
bash tools/r03_pmc.sh $tag || exit 1
