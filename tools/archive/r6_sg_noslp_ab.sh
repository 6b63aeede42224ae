for rep in 1 2; do
for v in product variant; do
  if [ $v = variant ]; then export PYTV4D_LIB=$PWD/pytv-4d_amd/pytv/libpytv4d_hip_sgnoslp.so; else unset PYTV4D_LIB; fi
  echo "== $v rep $rep"; python3 tools/op_bench.py 64x8x1024x1024 2>/dev/null | grep "tv_subgrad_fused "
done
done
