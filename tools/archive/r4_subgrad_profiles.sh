#!/bin/bash
# round 4: rocprofv3 kernel stats + SQ / FETCH / WRITE digests of the one-pass sub-gradient, round-3 kernel and the round-4 experiment
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
for k in 2 3; do for s in hybrid upwind; do
  export TV_SG_KERNEL=$k
  OPS=tv_subgrad_fused bash tools/prof_op.sh r4sg_k${k}_$s 64x8x1024x1024 $s > gpurun_out/prof_r4sg_k${k}_$s.log 2>&1
  cp gpurun_out/op_r4sg_k${k}_$s/digest.json gpurun_out/r4_subgrad_k${k}_${s}_sq_pmc_digest.json
  find gpurun_out/op_r4sg_k${k}_$s/trace -name "*kernel_stats.csv" -exec cp {} gpurun_out/r4_subgrad_k${k}_${s}_kernel_stats.csv \;
done; done
ls -la gpurun_out/r4_subgrad_*
