#!/bin/bash
# usage (on the GPU box): VARIANTS="y1 y2" bash tools/st_ab.sh -- streaming kernels (tv_D, tv_normal_op): time + FETCH_SIZE per variant library
R=$GRAFT_REPO_ROOT
cd $R
for v in base ${VARIANTS}; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  echo "== $v"
  OPS=tv_D,tv_normal_op python3 tools/op_bench.py 256x8x1024x1024 hybrid upwind central 2>&1 | grep "tv_"
done
for v in ${VARIANTS}; do
  VARIANT=$v SCRIPT="tools/op_bench.py 256x8x1024x1024 hybrid central" KFILTER=stream OPS=tv_D,tv_normal_op bash tools/fetch_ab.sh 2>&1 | grep -v "^base" 
done
VARIANT=${VARIANTS%% *} SCRIPT="tools/op_bench.py 256x8x1024x1024 hybrid central" KFILTER=stream OPS=tv_D,tv_normal_op bash tools/fetch_ab.sh 2>&1 | grep "^base"
