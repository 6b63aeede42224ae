#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_northstar_arena_first_command.json 2> gpurun_out/r5_bench_arena_err.txt
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_northstar_arena_first_command.json'))
print('ms_per_step', d['ms_per_step'], 'sweep', d['roofline']['ms_per_launch'], 'frac', d['roofline']['frac'], 'fixup', d['roofline_fixup']['ms_per_launch'], 'traffic', d['roofline']['traffic'], d['roofline_fixup']['traffic'])
print('placement', d['placement_tuning']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])
PY
tail -3 gpurun_out/r5_bench_arena_err.txt
for i in 1 2; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --pmc off 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run', d['ms_per_step'], d['roofline']['ms_per_launch'], d['placement_tuning'])"; done
timeout 1500 python -m pytest tests/test_gpu_cp_r4.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_pitch.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -6
