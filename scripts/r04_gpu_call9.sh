#!/bin/bash
# Round 4: the bench line as the driver takes it (default flags, then --steps 20 --warmup 5), smoke first.
set -o pipefail
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
echo "== bench default" ; timeout -k 10 900 python bench.py > gpurun_out/r04/bench_default_full.json 2> gpurun_out/r04/bench_default_full.err || { tail -20 gpurun_out/r04/bench_default_full.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_default_full.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms_per_step', d['ms_per_step'], 'frac', round(d['roofline']['frac'],4))
print('cpu', {k: d['cpu_baseline'].get(k) for k in ('value','cores','kind')})
for o in d['other_workloads']: print(' ', o.get('workload','')[:70], round(o.get('frames_per_s',0)), round(o.get('frac',0),4))
h=d['host_fed']; print('hot', round(h['compact8_zero_copy_frames_per_s']), round(h['aos40_copy_frames_per_s']))
for k in ('64x1','16x4'):
    v=h['config4_64_streams'][k]; print(k, round(v['frames_per_s_steady']), round(v['frames_per_s_wall']), v['setup_ms'], v['worker_time_share'])
PY
echo "== bench driver flags" ; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_flags.json 2> gpurun_out/r04/bench_driver_flags.err || { tail -20 gpurun_out/r04/bench_driver_flags.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r04/bench_driver_flags.json').read().strip().splitlines()[-1]); print('driver flags: value', round(d['value']), 'ms_per_step', d['ms_per_step'], 'frac', round(d['roofline']['frac'],4))"
echo "== PCIe read sweep" ; READBW_HOST=1 READBW_MB=64 timeout -k 10 120 scripts/micro/readbw > gpurun_out/r04/readbw_host.txt 2>&1; cat gpurun_out/r04/readbw_host.txt
