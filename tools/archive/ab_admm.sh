#!/bin/bash
# usage (on the GPU box): VARIANT=r2a [ROUNDS=2] bash tools/ab_admm.sh -- ADMM outer iteration (tools/admm_bench.py), default library vs a variant
R=$GRAFT_REPO_ROOT
for r in $(seq ${ROUNDS:-2}); do for v in base ${VARIANT}; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  python3 $R/tools/admm_bench.py 2>&1 | grep "single-reduction" | cut -c1-60 | sed "s/^/$v  /"
done; done
