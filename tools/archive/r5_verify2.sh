#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash tools/r5_al_ab.sh
python3 bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 > gpurun_out/r5_bench_admm_config4slab_upwind.json 2> gpurun_out/r5_bench_admm_err.txt
tail -c 2500 gpurun_out/r5_bench_admm_config4slab_upwind.json; tail -3 gpurun_out/r5_bench_admm_err.txt
timeout 2400 python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -25 > gpurun_out/r5_verify2_rccl.txt
cat gpurun_out/r5_verify2_rccl.txt
