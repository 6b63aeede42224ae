#!/bin/bash
# round 5: first GPU pass over what this round added (tests + the ADMM bench line as a first command)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 bench.py --solver admm --workload config4-slab --scheme upwind --steps 10 --warmup 3 > gpurun_out/r5_bench_admm_config4slab_upwind.json 2> gpurun_out/r5_bench_admm_err.txt
tail -c 3000 gpurun_out/r5_bench_admm_config4slab_upwind.json
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15 > gpurun_out/r5_verify1_fullsize.txt
timeout 2400 python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -25 > gpurun_out/r5_verify1_rccl.txt
cat gpurun_out/r5_verify1_fullsize.txt gpurun_out/r5_verify1_rccl.txt; tail -5 gpurun_out/r5_bench_admm_err.txt
