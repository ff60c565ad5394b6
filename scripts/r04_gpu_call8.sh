#!/bin/bash
# Round 4: full GPU suite incl. the run-pattern tests, a 2-minute soak, then the profile pass on the final kernels.
set -o pipefail
mkdir -p gpurun_out/r04
echo "== gpu tests" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite_final.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite_final.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite_final.log
echo "== soak 150 s" ; MTGPU_SOAK_SECONDS=150 MTGPU_SOAK_SEED=4004 timeout -k 10 400 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -s > gpurun_out/r04/soak.log 2>&1 || { tail -40 gpurun_out/r04/soak.log; exit 1; }
grep -E "soak:|passed|failed" gpurun_out/r04/soak.log
echo "== profiles" ; timeout -k 10 900 bash scripts/profile_r04.sh > gpurun_out/r04/profile_final.log 2>&1 || { tail -30 gpurun_out/r04/profile_final.log; exit 1; }
grep -E "scan_frames" gpurun_out/r04/profile_final.log | cut -c1-60,200-300
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_pmc_traffic.json'))
for k,v in d.items(): print(k, 'kernel_ms', round(v['kernel_ms_in_stats_run'],4), 'ratio', round(v['traffic_over_algorithmic'],4), 'frac', round(v['algorithmic_bytes_per_launch']/v['kernel_ms_in_stats_run']/1e6/8000,4))
PY
