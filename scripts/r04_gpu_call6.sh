#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
echo "== A/B" ; AB_ROUNDS=2 ONLY=0,1,2,4,7 bash scripts/ab_libs_r04.sh base runs runs3 runs2 > gpurun_out/r04/ab_pan2.log 2>&1 || { tail -30 gpurun_out/r04/ab_pan2.log; exit 1; }
grep -E "^==|^case" gpurun_out/r04/ab_pan2.log
echo "== gpu tests (runs2 build)" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite_runs2.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite_runs2.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite_runs2.log
