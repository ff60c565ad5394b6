#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
echo "== gpu tests" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite_14.log 2>&1 || { tail -60 gpurun_out/r04/gpu_suite_14.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite_14.log
echo "== soak 600 s" ; MTGPU_SOAK_SECONDS=600 MTGPU_SOAK_SEED=20261004 timeout -k 10 800 python -m pytest tests/test_gpu_soak.py -x -q -m gpu -s > gpurun_out/r04/soak3.log 2>&1 || { tail -40 gpurun_out/r04/soak3.log; exit 1; }
grep -E "soak:|passed|failed" gpurun_out/r04/soak3.log
