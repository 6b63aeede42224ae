#!/bin/bash
# Round 6: tv_subgrad.hip / tv_subgrad_norms.hip / tv_sgstep.hip compiled with -fno-slp-vectorize (variant library in PYTV4D_LIB) against the product,
# interleaved on one box: the operator at 64x8x1024x1024 and the descent loop at the north star.
VAR=$PWD/pytv-4d_amd/pytv/libpytv4d_hip_sgnoslp.so
for rep in 1 2 3; do
for v in product variant; do
  if [ $v = variant ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
  echo "== $v rep $rep"; python3 tools/op_bench.py 64x8x1024x1024 2>/dev/null | grep "tv_subgrad_fused"
  SG_TUNE=0 python3 tools/sg_bench.py 256x8x1024x1024 hybrid upwind downwind central 2>/dev/null | grep "one-pass"
done
done
