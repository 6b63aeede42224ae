#!/bin/bash
# round 5: interleaved A/B of the ALIGNED-tile (ring slot) build of the one-pass sub-gradient kernel against the round-3 tiles in the
# SAME library (TV_SG_ALIGNED=1 / 0), one box: single operators at 64x8x1024x1024 and the descent loop at 256x8x1024x1024
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/r5_al_ab.txt; : > $O
for r in 1 2; do for v in 1 0; do
  echo "== TV_SG_ALIGNED=$v (round $r)" >> $O
  TV_SG_ALIGNED=$v python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind downwind central 2>&1 | grep -i "subgrad_fused" >> $O
done; done
for v in 1 0; do
  echo "== descent loop, TV_SG_ALIGNED=$v" >> $O
  TV_SG_ALIGNED=$v SG_TUNE=0 python3 tools/sg_bench.py 256x8x1024x1024 hybrid upwind central 2>&1 | grep "one-pass" >> $O
done
cat $O
