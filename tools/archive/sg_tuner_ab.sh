#!/bin/bash
# the descent loop on the north-star volume with and without the placement tuner of SubgradientDescent, interleaved in one lease
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in 1 2 3; do for t in 0 1; do
  echo "== SG_TUNE=$t"; SG_TUNE=$t python3 $R/tools/sg_bench.py 256x8x1024x1024 ${SCHEMES:-hybrid upwind} 2>&1 | grep -A1 one-pass | cut -c1-400
done; done
