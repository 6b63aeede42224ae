# usage (GPU box): VARIANTS="a b c" OPS=<regex> bash tools/ab_lib2.sh <shape> <scheme...>   -- tools/op_bench.py under each build of the library
R=$GRAFT_REPO_ROOT
for r in 1 2; do for v in base $VARIANTS; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  echo "== $v"; python3 $R/tools/op_bench.py "$@" 2>&1 | grep -E "${OPS:-tv_subgrad_fused }"
done; done
