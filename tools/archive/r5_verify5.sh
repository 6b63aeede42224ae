#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
timeout 2400 python -m pytest tests/test_gpu_admm_fused.py tests/test_gpu_admm_ops.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_pitch.py tests/test_gpu_cp_r4.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -8
for s in upwind hybrid central; do python3 bench.py --solver admm --workload config4-slab --scheme $s --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5b_bench_admm_config4slab_$s.json 2>/dev/null; python3 -c "
import json,sys; d=json.loads([l for l in open('gpurun_out/r5b_bench_admm_config4slab_$s.json').read().splitlines() if l.startswith('{')][-1]); print('$s', d['ms_per_step'], 'sweep', d['roofline']['ms_per_launch'], round(d['roofline']['frac'],3), 'traffic', d['roofline']['traffic'], d['roofline']['bytes_per_launch'], 'xsolve', d['roofline_xsolve']['ms_per_outer_iteration'], 'words', d['words_per_voxel_and_outer_iteration'], d['loss_first_last'])"; done
