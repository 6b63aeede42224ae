#!/bin/bash
# interleaved A/B of the one-pass sub-gradient kernels: TV_SG_KERNEL=3 (round 4: a lane = 2 rows x 2 columns) against 2 (round 3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in 1 2; do for k in 2 3; do
  echo "== TV_SG_KERNEL=$k: single operators, ${SHAPE:-64x8x1024x1024}"; TV_SG_KERNEL=$k python3 $R/tools/op_bench.py ${SHAPE:-64x8x1024x1024} ${SCHEMES:-hybrid upwind downwind central} 2>&1 | grep -i "subgrad_fused"
  echo "== TV_SG_KERNEL=$k: descent loop, 256x8x1024x1024"; TV_SG_KERNEL=$k python3 $R/tools/sg_bench.py 256x8x1024x1024 ${SCHEMES:-hybrid upwind downwind central} 2>&1 | grep one-pass
done; done
