#!/bin/bash
# Round 4, GPU call 4: per-channel L2 <-> fabric counters of several allocations of the north-star state inside one process,
# the pitch benchmark, the ADMM x-solve study, the round's new tests.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call4
mkdir -p "$O"
export TMPDIR=/tmp
for C in "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ" "TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ"; do
  T=$(echo $C | tr ' ' '+')
  D=/tmp/r4pc_$T
  rm -rf $D
  ( cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format json -d $D -o p -- python3 $R/tools/placement_retry.py 4 > $O/retry_pmc_$T.txt 2> $O/retry_pmc_$T.err )
  ls -la $(find $D -name "*.json" | head -3) > $O/json_$T.ls 2>&1
  python3 tools/pmc_channels.py $D > $O/channels_$T.txt 2> $O/channels_$T.err; tail -40 $O/channels_$T.txt | cut -c1-220
done
timeout 900 python3 tools/pitch_bench.py > $O/pitch_bench.txt 2> $O/pitch_bench.err; cat $O/pitch_bench.txt | cut -c1-330; tail -3 $O/pitch_bench.err
timeout 900 python3 tools/archive/admm_xsolve_study.py 32x16x1024x1024 50 > $O/admm_study_f32.txt 2> $O/admm_study.err; cat $O/admm_study_f32.txt | cut -c1-330
timeout 900 python3 tools/archive/admm_xsolve_study.py 16x16x1024x1024 50 --f64 > $O/admm_study_f64.txt 2>> $O/admm_study.err; cat $O/admm_study_f64.txt | cut -c1-330
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_admm_fused.py tests/test_gpu_rccl.py tests/test_gpu_pitch.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
