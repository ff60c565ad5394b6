#!/bin/bash
# Round 4, third GPU call: pipe set-up probe, GPU suite, profiles of the vote-heavy banded case, compact affine probe.
set -o pipefail
mkdir -p gpurun_out/r04
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"
echo "== pin probe" ; timeout -k 10 200 scripts/micro/pin_probe > gpurun_out/r04/pin_probe2.json || exit 1
python -c "import json; d=json.load(open('gpurun_out/r04/pin_probe2.json')); print(json.dumps(d['concurrent_pipe_shaped_setup']))"
echo "== gpu tests" ; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_suite.log 2>&1 || { tail -40 gpurun_out/r04/gpu_suite.log; exit 1; }
tail -3 gpurun_out/r04/gpu_suite.log
echo "== profiles" ; timeout -k 10 900 bash scripts/profile_r04.sh > gpurun_out/r04/profile.log 2>&1 || { tail -30 gpurun_out/r04/profile.log; exit 1; }
tail -12 gpurun_out/r04/profile.log
echo "== compact affine" ; timeout -k 10 600 python scripts/compact_affine_r04.py > gpurun_out/r04/compact_affine.json 2> gpurun_out/r04/compact_affine.log || { tail -20 gpurun_out/r04/compact_affine.log; exit 1; }
cat gpurun_out/r04/compact_affine.json
