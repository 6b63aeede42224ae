#!/bin/bash
# round 5: interleaved A/B of the ALIGNED-tile (ring slot) build of the one-pass sub-gradient kernel against the round-3 tiles in ONE
# library built with AL for all four schemes (TV_VARIANT=alall TV_EXTRA_FLAGS=-DTV_SG2_AL=2 python3 pytv-4d_amd/build.py), one box:
# TV_SG_ALIGNED=2 (aligned tiles wherever instantiated) / 0 (round-3 tiles); single operators at 64x8x1024x1024, descent loop at 256 planes
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export PYTV4D_LIB=$PWD/pytv-4d_amd/pytv/libpytv4d_hip_alall.so
O=gpurun_out/${OUT:-r5_al_ab2.txt}; : > $O
for r in 1 2; do for v in 2 0; do
  echo "== TV_SG_ALIGNED=$v (round $r)" >> $O
  TV_SG_ALIGNED=$v python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind downwind central 2>&1 | grep -i "subgrad_fused" >> $O
done; done
for r in 1 2; do for v in 2 0; do
  echo "== descent loop, TV_SG_ALIGNED=$v (round $r)" >> $O
  TV_SG_ALIGNED=$v SG_TUNE=0 python3 tools/sg_bench.py 256x8x1024x1024 hybrid upwind downwind central 2>&1 | grep "one-pass" >> $O
done; done
cat $O
