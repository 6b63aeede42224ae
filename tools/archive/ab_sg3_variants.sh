R=$GRAFT_REPO_ROOT
for r in 1 2; do for v in base sg3st sg3nox; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  echo "== $v"; python3 $R/tools/op_bench.py 64x8x1024x1024 hybrid upwind central 2>&1 | grep -i "subgrad_fused"
  python3 $R/tools/sg_bench.py 256x8x1024x1024 hybrid upwind 2>&1 | grep one-pass
done; done
