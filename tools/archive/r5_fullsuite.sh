#!/bin/bash
# the driver's round-end check, run by the builder: the whole GPU suite on the product library
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
timeout 3300 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r5_fullsuite.txt
cat gpurun_out/r5_fullsuite.txt
