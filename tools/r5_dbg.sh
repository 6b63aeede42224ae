#!/bin/bash
# chunk-length sensitivity of every z-chunked kernel: op_bench with TV_ZCHUNK = 0 (each kernel's own rule) / 16 / 32, and the ADMM / CP lines
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
for zc in 0 16 32; do
  echo "=== TV_ZCHUNK=$zc op_bench 64x8x1024x1024"
  TV_ZCHUNK=$zc python3 tools/op_bench.py 64x8x1024x1024 2>&1 | grep -E "^(hybrid|upwind|central) " | cut -c1-100
done
for zc in 0 32; do
  for s in upwind hybrid; do
    TV_ZCHUNK=$zc python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --no-cpu-baseline --pmc off > $O/tmp_zc.json 2>/dev/null
    python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_zc.json').read().splitlines() if l.startswith('{')][-1]); print('admm TV_ZCHUNK=$zc $s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), 'fixup', round(d['roofline_fixup']['ms_per_launch'],3), 'xsolve', round(d['roofline_xsolve']['ms_per_outer_iteration'],3), d['loss_first_last'])"
  done
done
