#!/bin/bash
# PFX A/B without the placement lottery: the CP state carved out of ONE allocation (tools/archive/slab_placement_probe.py, layout slab+32), 4 constructions
# per run, product library and libpytv4d_hip_pfx.so interleaved three times on one box; sweep ms of the north-star volume (hybrid)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
VAR=$R/pytv-4d_amd/pytv/libpytv4d_hip_pfx.so
for rep in 1 2 3; do
  for lib in product pfx; do
    if [ $lib = pfx ]; then export PYTV4D_LIB=$VAR; else unset PYTV4D_LIB; fi
    echo "== $lib (rep $rep)"; python3 tools/archive/slab_placement_probe.py 256x8x1024x1024 4 slab+32,separate 2>&1 | grep -E "^slab|^separate" | cut -c1-200
  done
done
