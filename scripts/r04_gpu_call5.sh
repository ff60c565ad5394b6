#!/bin/bash
# Round 4, fifth GPU call: parity suite on the run-aggregated vote path, then base vs runs builds interleaved.
set -o pipefail
mkdir -p gpurun_out/r04
echo "== gpu tests (runs build)" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite_runs.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite_runs.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite_runs.log
echo "== A/B" ; AB_ROUNDS=2 bash scripts/ab_libs_r04.sh base runs > gpurun_out/r04/ab_pan.log 2>&1 || { tail -30 gpurun_out/r04/ab_pan.log; exit 1; }
grep -E "^==|^case" gpurun_out/r04/ab_pan.log
