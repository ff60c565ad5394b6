#!/bin/bash
# Round 4: per-workgroup phase times of the vote-heavy banded case, round-3 vote path vs run-aggregated (instrumented
# builds), then the batch-size A/B of the host feed on the final host layer.
set -o pipefail
mkdir -p gpurun_out/r04
P=motion-estimated-video-trimmer_amd
cp $P/libmtgpu.so /tmp/libmtgpu_keep.so
for v in base_pt runs_pt; do
  cp $P/libmtgpu_$v.so $P/libmtgpu.so
  echo "== $v"
  AB_PAN=1 timeout -k 10 300 python scripts/phase_times.py 4k_fine_dense4 256 4 1 4k_fine 512 4 1 || exit 1
done > gpurun_out/r04/phase_times_pan.txt 2>&1
cp /tmp/libmtgpu_keep.so $P/libmtgpu.so
cat gpurun_out/r04/phase_times_pan.txt
echo "== host feed batch sizes" ; SET=batch PASSES=2 REPS=600 timeout -k 10 700 python scripts/host_feed_ab_r04.py > gpurun_out/r04/host_feed_ab_batch.json 2> gpurun_out/r04/host_feed_ab_batch.log || { tail -20 gpurun_out/r04/host_feed_ab_batch.log; exit 1; }
cat gpurun_out/r04/host_feed_ab_batch.log
