#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_admm_fused.py tests/test_gpu_admm_ops.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for s in upwind central; do
  python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --no-cpu-baseline --pmc off > $O/tmp_x.json 2>/dev/null
  python3 -c "
import json; d=json.loads([l for l in open('$O/tmp_x.json').read().splitlines() if l.startswith('{')][-1]); print('$s', round(d['ms_per_step'],3), 'sweep', round(d['roofline']['ms_per_launch'],3), 'xsolve', round(d['roofline_xsolve']['ms_per_outer_iteration'],3), d['loss_first_last'])"
done
