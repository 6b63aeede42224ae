#!/bin/bash
# A/B on one box: (1) config1 / config2, round-3 tree against the current tree, interleaved; (2) window-of-4 + 16 waves per CU
# variant (libpytv4d_hip_twn4.so) against the default library with tools/op_bench.py.
# Prerequisites (authoring container, before gpurun):
#   git worktree add r3tree 93a5ad8 && (cd r3tree && python pytv-4d_amd/build.py)
#   TV_VARIANT=twn4 TV_EXTRA_FLAGS="-DTV_TWN=4 -DTV_WAVES=4" python pytv-4d_amd/build.py
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R" || exit 1
O=$R/gpurun_out/r4call11
mkdir -p "$O"
export TMPDIR=/tmp
for rep in 1 2; do
 for w in config2 config1; do
  (cd r3tree && python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --pmc off) > $O/r3_${w}_$rep.json 2>/dev/null
  python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --pmc off > $O/r4_${w}_$rep.json 2>/dev/null
 done
done
python3 - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "r[34]_config*.json"))):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), 'ms/step %.3f sweep %.3f fixup %.3f' % (d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline_fixup']['ms_per_launch']))
    except Exception as e:
        print(f, 'unreadable', e)
PY
for rep in 1 2; do
  python3 tools/op_bench.py 64x8x1024x1024 > $O/op_default_$rep.txt 2>&1
  PYTV4D_LIB=$R/pytv-4d_amd/pytv/libpytv4d_hip_twn4.so python3 tools/op_bench.py 64x8x1024x1024 > $O/op_twn4_$rep.txt 2>&1
done
python3 - $O <<'PY'
import sys,os
def load(f):
    d={}
    for line in open(f):
        p=line.split()
        if len(p)>=5 and p[0] in ("hybrid","upwind","downwind","central"):
            d[(p[0],p[1])]=float(p[2])
    return d
o=sys.argv[1]
a=[load(os.path.join(o,"op_default_%d.txt"%k)) for k in (1,2)]
b=[load(os.path.join(o,"op_twn4_%d.txt"%k)) for k in (1,2)]
print("%-9s %-24s default ms (2 runs)   window-of-4 + 4 waves/SIMD ms (2 runs)" % ("scheme","op"))
for key in a[0]:
    if all(key in x for x in a+b):
        print("%-9s %-24s %7.3f %7.3f      %7.3f %7.3f" % (key[0],key[1],a[0][key],a[1][key],b[0][key],b[1][key]))
PY
