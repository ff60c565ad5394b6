#!/bin/bash
# Developer tool: alternate builds of libmtgpu.so (libmtgpu_<name>.so beside it) over scripts/ab_pan_r04.py.
# Usage: bash scripts/ab_libs_r04.sh base runs   (AB_ROUNDS rounds; restores the LAST named build as libmtgpu.so)
P=motion-estimated-video-trimmer_amd
for r in $(seq 1 ${AB_ROUNDS:-2}); do
  for v in "$@"; do
    cp $P/libmtgpu_$v.so $P/libmtgpu.so
    echo "== $v round $r"
    timeout -k 10 400 python scripts/ab_pan_r04.py || exit 1
  done
done
