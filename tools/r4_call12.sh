#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call12
mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_pitch.py -x -q -m gpu > $O/pytest_pitch.txt 2>&1; tail -8 $O/pytest_pitch.txt
timeout 900 python3 tools/pitch_bench.py 64x8x1001x1001 64x8x1000x1000 > $O/pitch_bench.txt 2>&1; cat $O/pitch_bench.txt | cut -c1-330
timeout 900 python3 tools/stress_ops.py > $O/stress_ops.txt 2>&1; tail -3 $O/stress_ops.txt
