#!/bin/bash
# Round 4, third diagnosis lease: clocks / power DURING the iteration in the first and in a later process of the lease, the
# distance test, then the parity tests of the round's kernels.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4diag3
mkdir -p "$O"
export TMPDIR=/tmp
uptime > $O/uptime.txt; cat /proc/uptime >> $O/uptime.txt
python3 tools/clock_trace.py 3 > $O/clock_trace_first.txt 2> $O/ct1.err; head -3 $O/clock_trace_first.txt | cut -c1-400; sed -n 4,12p $O/clock_trace_first.txt
python3 tools/clock_trace.py 3 > $O/clock_trace_second.txt 2> $O/ct2.err; head -3 $O/clock_trace_second.txt | cut -c1-400; sed -n 4,12p $O/clock_trace_second.txt
timeout 300 tools/deltatest 4 > $O/deltatest.txt 2>&1; cat $O/deltatest.txt | head -80
python3 tools/clock_trace.py 3 > $O/clock_trace_third.txt 2> $O/ct3.err; head -2 $O/clock_trace_third.txt | cut -c1-400
timeout 1200 python3 -m pytest tests/test_gpu_pitch.py -x -q -m gpu > $O/pytest_pitch.txt 2>&1; tail -5 $O/pytest_pitch.txt
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_admm_fused.py -x -q -m gpu > $O/pytest_fullsize.txt 2>&1; tail -15 $O/pytest_fullsize.txt
