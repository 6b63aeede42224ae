#!/bin/bash
# round 5: the aligned tiles for fp64 upwind / downwind -- parity tests, then an interleaved A/B (TV_SG_ALIGNED=1 / 0) of the single operators at
# 32x8x1024x1024 fp64 and of the descent loop at 128x8x1024x1024 fp64
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/r5_f64_aligned_ab.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_subgrad_onepass.py -x -q 2>&1 | tail -4 >> $O
for r in 1 2; do for v in 1 0; do
  echo "== fp64, TV_SG_ALIGNED=$v (round $r)" >> $O
  DTYPE=f64 TV_SG_ALIGNED=$v python3 tools/op_bench.py 32x8x1024x1024 upwind downwind 2>&1 | grep -i "subgrad_fused" >> $O
done; done
cat $O
