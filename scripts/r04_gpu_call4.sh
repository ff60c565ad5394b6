#!/bin/bash
# Round 4, fourth GPU call: GPU suite on the stream-pool build, then the final host layer against each of its parts
# switched back (interleaved, 2 passes, the bench's own run length).
set -o pipefail
mkdir -p gpurun_out/r04
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"
echo "== gpu tests" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite.log
echo "== host feed A/B (final)" ; SET=final PASSES=${PASSES:-2} REPS=${REPS:-600} timeout -k 10 1000 python scripts/host_feed_ab_r04.py > gpurun_out/r04/host_feed_ab3.json 2> gpurun_out/r04/host_feed_ab3.log || { tail -20 gpurun_out/r04/host_feed_ab3.log; exit 1; }
cat gpurun_out/r04/host_feed_ab3.log
