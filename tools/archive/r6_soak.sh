#!/bin/bash
# Round 6, final library: the stress tools with seeds they have not run with before.  Every tool under its own timeout.
O=gpurun_out
run() { name=$1; shift; echo "== $name: $*"; timeout 600 "$@" 2>&1 | grep -v "amdgpu.ids" | tail -4; echo "rc=${PIPESTATUS[0]}"; }
STRESS_SEED=7 run stress_small python3 tools/stress_small.py 400
STRESS_SEED=8 run stress_small_b python3 tools/stress_small.py 200
STRESS_SEED=5 run stress_ops python3 tools/stress_ops.py 300
STRESS_SEED=11 run stress_subgrad python3 tools/stress_subgrad.py 300
STRESS_SEED=3 run stress_fused python3 tools/stress_fused.py
STRESS_SEED=23 run stress_multirank python3 tools/stress_multirank.py 12
