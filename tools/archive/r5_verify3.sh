#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
timeout 1300 python3 tools/archive/slab_placement_probe.py 256x8x1024x1024 4 slab+16,slab+32,slab+8,slab+24,slab+48,slab+64,slab+32,slab+40,separate > gpurun_out/r5_slab_placement_probe3.txt 2>&1; cat gpurun_out/r5_slab_placement_probe3.txt
timeout 1200 python -m pytest tests/test_gpu_cp_r4.py tests/test_gpu_configs.py -x -q 2>&1 | tail -6 > gpurun_out/r5_verify3_cp.txt; cat gpurun_out/r5_verify3_cp.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off > gpurun_out/r5_bench_northstar_fid_both.json 2>/dev/null; head -c 400 gpurun_out/r5_bench_northstar_fid_both.json; echo
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_northstar_fid_both.json'))
print('ms_per_step', d['ms_per_step'], 'sweep', d['roofline']['ms_per_launch'], 'fixup', d['roofline_fixup']['ms_per_launch'], 'series', d['series_ms']['step'])
PY
