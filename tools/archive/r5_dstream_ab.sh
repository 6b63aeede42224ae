#!/bin/bash
# round 5, late: tv_D's streaming kernel after the straight-line rewrite (tv_dstream.h), one box: the parity tests that exercise tv_D, then
# tools/op_bench.py with this library and with the library of the commit before (PYTV4D_LIB=.../libpytv4d_hip_base.so), twice
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pitch.py tests/test_gpu_configs.py tests/test_gpu_weight_volume.py tests/test_gpu_multirank.py tests/test_gpu_admm_ops.py -x -q 2>&1 | tail -5
BASE=$R/pytv-4d_amd/pytv/libpytv4d_hip_base.so
for rep in 1 2; do
  for lib in new base; do
    [ $lib = base ] && [ ! -f $BASE ] && continue
    if [ $lib = base ]; then export PYTV4D_LIB=$BASE; else unset PYTV4D_LIB; fi
    echo "--- op_bench $lib 64x8x1024x1024"; python3 tools/op_bench.py 64x8x1024x1024 2>&1 | grep -E "tv_D |tv_normal_op|tv_cheb" | cut -c1-80
    echo "--- op_bench $lib 32x16x1024x1024"; python3 tools/op_bench.py 32x16x1024x1024 2>&1 | grep -E "tv_D |tv_normal_op|tv_cheb" | cut -c1-80
  done
done
unset PYTV4D_LIB
