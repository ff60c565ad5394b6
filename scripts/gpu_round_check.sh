#!/bin/bash
# What a round runs on the GPU box before it is handed in (one gpurun call, ~8 minutes):
#   the whole -m gpu suite, a soak (MTGPU_SOAK_SECONDS, default 150), the smoke test, the bench line with the driver's flags.
set -o pipefail
mkdir -p gpurun_out/check
echo "== gpu tests" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/check/gpu_suite.log 2>&1 || { tail -60 gpurun_out/check/gpu_suite.log; exit 1; }
tail -3 gpurun_out/check/gpu_suite.log
echo "== soak" ; MTGPU_SOAK_SECONDS=${MTGPU_SOAK_SECONDS:-150} MTGPU_SOAK_SEED=${MTGPU_SOAK_SEED:-$RANDOM} timeout -k 10 900 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -s > gpurun_out/check/soak.log 2>&1 || { tail -40 gpurun_out/check/soak.log; exit 1; }
grep -E "soak:|passed|failed" gpurun_out/check/soak.log
echo "== smoke" ; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
echo "== bench (driver flags)" ; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/check/bench_driver_flags.json 2> gpurun_out/check/bench_driver_flags.err || { tail -20 gpurun_out/check/bench_driver_flags.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/check/bench_driver_flags.json').read().strip().splitlines()[-1]); print('value', round(d['value']), 'ms_per_step', d['ms_per_step'], 'frac', round(d['roofline']['frac'],4))"
