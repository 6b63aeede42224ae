#!/bin/bash
# usage (on the GPU box): VARIANTS="nc" OPS=tv_normal_op [ROUNDS=3] [SHAPE=256x8x1024x1024] [SCHEMES="hybrid central"] bash tools/ab_ops.sh
# interleaved timing A/B of tools/op_bench.py with the default library and variant builds
R=$GRAFT_REPO_ROOT
for r in $(seq ${ROUNDS:-3}); do for v in base ${VARIANTS}; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  python3 $R/tools/op_bench.py ${SHAPE:-256x8x1024x1024} ${SCHEMES:-hybrid central} 2>&1 | grep "tv_" | sed "s/^/$v  /"
done; done
