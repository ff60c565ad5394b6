#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04
echo "== gpu tests" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite_10.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite_10.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite_10.log
echo "== hot stream A/B" ; timeout -k 10 900 python scripts/host_hot_ab_r04.py > gpurun_out/r04/host_hot_ab.json 2> gpurun_out/r04/host_hot_ab.log || { tail -20 gpurun_out/r04/host_hot_ab.log; exit 1; }
cat gpurun_out/r04/host_hot_ab.log
