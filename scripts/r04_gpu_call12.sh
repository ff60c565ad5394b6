#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
echo "== bench default" ; T0=$(date +%s); timeout -k 10 900 python bench.py > gpurun_out/r04/bench_default_full.json 2> gpurun_out/r04/bench_default_full.err || { tail -20 gpurun_out/r04/bench_default_full.err; exit 1; }
echo "bench.py wall: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default_full.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms_per_step', d['ms_per_step'], 'frac', round(d['roofline']['frac'],4))
print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')})
h=d['host_fed']; print('hot', round(h['compact8_zero_copy_frames_per_s']), round(h['aos40_copy_frames_per_s']), h['pcie_GBps'])
for k in ('64x1','16x4'):
    v=h['config4_64_streams'][k]; print(k, round(v['frames_per_s_steady']), round(v['frames_per_s_wall']), v['setup_ms'], {a: round(b,2) for a,b in v['worker_time_share'].items()}, v.get('cpu_gate'))
PY
echo "== soak 300 s" ; MTGPU_SOAK_SECONDS=300 MTGPU_SOAK_SEED=777 timeout -k 10 500 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -s > gpurun_out/r04/soak2.log 2>&1 || { tail -40 gpurun_out/r04/soak2.log; exit 1; }
grep -E "soak:|passed|failed" gpurun_out/r04/soak2.log
