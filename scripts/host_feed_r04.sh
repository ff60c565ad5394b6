#!/bin/bash
# Round 4, first GPU call: where does the host-fed path's time go?  (run on the GPU box)
set -o pipefail
mkdir -p gpurun_out/r04
echo "== pack_bench" ; timeout -k 10 400 scripts/micro/pack_bench --threads 1,4,16 --mb 512 > gpurun_out/r04/pack_bench_pinned.json || exit 1
cat gpurun_out/r04/pack_bench_pinned.json
echo "== pin_probe" ; timeout -k 10 200 scripts/micro/pin_probe > gpurun_out/r04/pin_probe.json || exit 1
cat gpurun_out/r04/pin_probe.json
echo "== host feed A/B" ; timeout -k 10 900 python scripts/host_feed_ab_r04.py > gpurun_out/r04/host_feed_ab.json 2> gpurun_out/r04/host_feed_ab.log || { tail -20 gpurun_out/r04/host_feed_ab.log; exit 1; }
cat gpurun_out/r04/host_feed_ab.log
