#!/bin/bash
# Round 4, second diagnosis lease: (1) placement retry inside the FIRST process of the lease, (2) physical make-up of allocations
# (tools/vmtest), (3) placement retry in a second process, (4) parity tests of this round.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4diag2
mkdir -p "$O"
export TMPDIR=/tmp
python3 tools/placement_retry.py 5 > $O/retry_first_process.txt 2> $O/retry_first.err; cat $O/retry_first_process.txt | cut -c1-260
timeout 600 tools/vmtest 8 > $O/vmtest.txt 2>&1; cat $O/vmtest.txt | cut -c1-220
python3 tools/placement_retry.py 5 > $O/retry_second_process.txt 2> $O/retry_second.err; cat $O/retry_second_process.txt | cut -c1-260
timeout 1200 python3 -m pytest tests/test_gpu_pitch.py -x -q -m gpu > $O/pytest_pitch.txt 2>&1; tail -5 $O/pytest_pitch.txt
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_admm_fused.py -x -q -m gpu > $O/pytest_fullsize.txt 2>&1; tail -15 $O/pytest_fullsize.txt
