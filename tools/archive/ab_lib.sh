#!/bin/bash
# usage (on the GPU box): VARIANT=ea [ROUNDS=2] [SCHEMES="hybrid upwind central"] bash tools/ab_lib.sh
# interleaved timing A/B of bench.py with the default library and a variant build
R=$GRAFT_REPO_ROOT
for r in $(seq ${ROUNDS:-2}); do for s in ${SCHEMES:-hybrid upwind central}; do for v in base ${VARIANT:-ea}; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  python3 $R/bench.py --pmc off --no-cpu-baseline --steps 10 --warmup 3 --scheme $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-5s %-8s ms/step %.3f sweep %.3f fixup %.3f' % ('$v','$s',d['ms_per_step'],d['roofline']['ms_per_launch'],d['roofline_fixup']['ms_per_launch']))"
done; done; done
