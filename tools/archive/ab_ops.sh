# A/B of two builds of the library over tools/op_bench.py (and the ADMM iteration): VARIANT=<suffix of libpytv4d_hip_<suffix>.so>
R=$GRAFT_REPO_ROOT
for r in 1 2; do for v in base ${VARIANT:-nont}; do
  if [ $v != base ]; then export PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_$v.so; else unset PYTV4D_LIB; fi
  echo "== $v"; python3 $R/tools/op_bench.py ${SHAPE:-64x8x1024x1024} ${SCHEMES:-hybrid central} 2>&1 | grep -E "${OPS:-tv_D |tv_DT |tv_admm_zu|tv_DT_axpy|tv_cp_dual|tv_cp_primal}"
  python3 $R/tools/admm_bench.py 32x16x1024x1024 5 2>&1 | tail -3
done; done
