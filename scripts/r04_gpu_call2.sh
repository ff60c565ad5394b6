#!/bin/bash
# Round 4, second GPU call: the whole GPU suite on the new host layer (lazy pinning, CPU gate, vector copy-out,
# offsets check), then the host-feed A/B with the gate and the throttling counters.
set -o pipefail
mkdir -p gpurun_out/r04
echo "== gpu tests" ; timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite.log
echo "== host feed A/B" ; PASSES=${PASSES:-2} REPS=${REPS:-400} timeout -k 10 1500 python scripts/host_feed_ab_r04.py > gpurun_out/r04/host_feed_ab2.json 2> gpurun_out/r04/host_feed_ab2.log || { tail -20 gpurun_out/r04/host_feed_ab2.log; exit 1; }
cat gpurun_out/r04/host_feed_ab2.log
