cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_subgrad_onepass.py -x -q 2>&1 | tail -15 > gpurun_out/r5_al_tests.txt
for v in 1 0; do echo "== TV_SG_ALIGNED=$v" >> gpurun_out/r5_al_ops.txt; TV_SG_ALIGNED=$v python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind downwind central 2>&1 | grep -i "subgrad_fused" >> gpurun_out/r5_al_ops.txt; done
for v in 1 0; do echo "== TV_SG_ALIGNED=$v" >> gpurun_out/r5_al_ops.txt; TV_SG_ALIGNED=$v python3 tools/op_bench.py 64x8x1024x1024 hybrid upwind downwind central 2>&1 | grep -i "subgrad_fused" >> gpurun_out/r5_al_ops.txt; done
cat gpurun_out/r5_al_tests.txt gpurun_out/r5_al_ops.txt
