#!/bin/bash
# interleaved A/B of the one-pass sub-gradient operators: the product library against TV_VARIANT builds (VARIANTS="a b ...")
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in 1 2; do for v in base ${VARIANTS:-stalign}; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  echo "== $v"; python3 $R/tools/op_bench.py ${SHAPE:-64x8x1024x1024} ${SCHEMES:-hybrid upwind central} 2>&1 | grep -i "subgrad_fused"
done; done
