#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call7
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python3 tools/pp_probe.py > $O/pp_probe.txt 2> $O/pp_probe.err; cat $O/pp_probe.txt | cut -c1-200; tail -3 $O/pp_probe.err
timeout 3000 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_pitch.py > $O/pytest_rest.txt 2>&1; tail -15 $O/pytest_rest.txt
timeout 600 python3 -m pytest tests/test_gpu_pitch.py -x -q -m gpu > $O/pytest_pitch.txt 2>&1; tail -5 $O/pytest_pitch.txt
