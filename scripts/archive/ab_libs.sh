#!/bin/bash
# Developer tool: alternate several builds of libmtgpu.so (libmtgpu_<name>.so next to it) over the same
# probe, AB_ROUNDS rounds.  Usage: bash scripts/ab_libs.sh old new ...   (restore libmtgpu.so afterwards)
P=motion-estimated-video-trimmer_amd
for r in $(seq 1 ${AB_ROUNDS:-3}); do
  for v in "$@"; do
    cp $P/libmtgpu_$v.so $P/libmtgpu.so
    echo "== $v round $r"
    timeout -k 10 200 python scripts/compact_probe.py || exit 1
  done
done
