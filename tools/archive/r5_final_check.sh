#!/bin/bash
# what the driver does at round end, run by the builder: bench as the first command, smoke(), the whole GPU suite
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 bench.py > gpurun_out/r5_bench_default_first_command.json 2> gpurun_out/r5_bench_default_err.txt
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default_first_command.json').read().splitlines() if l.startswith('{')][-1])
print('default bench: ms_per_step', d['ms_per_step'], 'it/s', d['value'], 'sweep', d['roofline']['ms_per_launch'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline_fixup']['traffic'], d['roofline']['traffic_source'][:90])
PY
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r5_fullsuite_final.txt; cat gpurun_out/r5_fullsuite_final.txt
