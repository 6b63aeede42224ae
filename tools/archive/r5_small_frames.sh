#!/bin/bash
# round 5, verdict item 8: the reference's own test shapes (pytv/tests.py:48 N = 100; README.md:76-79) -- how much of the small-frame
# gap is the z-chunk rule?  tools/pitch_bench.py 256x4x100x100 (and 20x4x100x100, the README shape) under TV_ZCHUNK = default / 2 / 4 / 8 / 16
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/r5_small_frames.txt; : > $O
for zc in 0 2 4 8 16; do
  echo "== TV_ZCHUNK=$zc" >> $O
  TV_ZCHUNK=$zc python3 tools/pitch_bench.py 256x4x100x100 20x4x100x100 2>&1 | grep -v "^$" >> $O
done
cat $O
