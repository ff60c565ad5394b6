#!/bin/bash
# Round 4, second GPU call: the whole GPU suite on the new host layer (lazy pinning, CPU gate, vector copy-out,
# offsets check), then the host-feed A/B with the gate and the throttling counters.
set -o pipefail
mkdir -p gpurun_out/r04
echo "== gpu tests" ; timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite.log
echo "== 6-rank rehearsal (gloo, same device: at most 6 processes may share the card on this pool; the 8-rank run is the driver's)"
timeout -k 10 400 python bench.py --gpus 6 --backend gloo --same-device --frames 1024 --steps 20 --warmup 3 --no-others --no-host --cpu-seconds 0 > gpurun_out/r04/rehearsal_6_ranks_gloo_same_device.json 2> gpurun_out/r04/rehearsal_6_ranks.log || { tail -20 gpurun_out/r04/rehearsal_6_ranks.log; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r04/rehearsal_6_ranks_gloo_same_device.json')); print('rehearsal: n_gpus', d['n_gpus'], 'ranks', len(d['ranks']), 'value', round(d['value']))"
echo "== host feed A/B" ; PASSES=${PASSES:-2} REPS=${REPS:-400} timeout -k 10 1500 python scripts/host_feed_ab_r04.py > gpurun_out/r04/host_feed_ab2.json 2> gpurun_out/r04/host_feed_ab2.log || { tail -20 gpurun_out/r04/host_feed_ab2.log; exit 1; }
cat gpurun_out/r04/host_feed_ab2.log
