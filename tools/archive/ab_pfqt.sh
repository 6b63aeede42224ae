# A/B on the GPU box: central dual prefetch in the windowed (M > 8) one-sweep kernels (build with -DTV_FUSED_PFQ_TWIN=0 as variant "nopfqt")
R=$GRAFT_REPO_ROOT
for r in 1 2; do for v in base nopfqt; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  echo "== $v"
  python3 $R/tools/ab_cp.py a= --rounds 2 --shape 64x16x1024x1024 --scheme central 2>&1 | tail -2
  ONLY=one-sweep python3 $R/tools/admm_bench.py 32x16x1024x1024 5 2>&1 | grep central
done; done
