#!/bin/bash
# Round-2 profile collection on the GPU box (run from the repo root): kernel-trace stats and, in
# SEPARATE passes, the FETCH_SIZE / WRITE_SIZE counters, for the headline workload and the banded
# 960x540 plan (VECTORS_NEEDED 4, the shipped env), plus kernel stats of the large merge.
# Summaries land in gpurun_out/r02_*; copy the ones to keep into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
HEAD="--steps 20 --warmup 3 --cpu-seconds 0 --no-others --no-host"
FINE="--workload 4k_fine --params shipped_env --frames 1024 --steps 10 --warmup 2 --cpu-seconds 0 --no-others --no-host"
run() { name=$1; shift; echo "== $name"; "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }; }

run r02_prof_1080p rocprofv3 --kernel-trace --stats -f csv -d $O/r02_prof_1080p -- python3 bench.py $HEAD
grep '^{' $O/r02_prof_1080p.log | tail -1 > $O/r02_bench_1080p_dense8x8.json
run r02_prof_fine rocprofv3 --kernel-trace --stats -f csv -d $O/r02_prof_fine -- python3 bench.py $FINE
grep '^{' $O/r02_prof_fine.log | tail -1 > $O/r02_bench_4k_fine_shipped_env.json
run r02_pmc_fetch_1080p rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/r02_pmc_fetch_1080p -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-others --no-host
run r02_pmc_write_1080p rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/r02_pmc_write_1080p -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-others --no-host
run r02_pmc_fetch_fine rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/r02_pmc_fetch_fine -- python3 bench.py --workload 4k_fine --params shipped_env --frames 1024 --steps 3 --warmup 1 --cpu-seconds 0 --no-others --no-host
run r02_pmc_write_fine rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/r02_pmc_write_fine -- python3 bench.py --workload 4k_fine --params shipped_env --frames 1024 --steps 3 --warmup 1 --cpu-seconds 0 --no-others --no-host
run r02_prof_merge rocprofv3 --kernel-trace --stats -f csv -d $O/r02_prof_merge -- python3 scripts/merge_rate.py
run r02_prof_compact rocprofv3 --kernel-trace --stats -f csv -d $O/r02_prof_compact -- python3 scripts/compact_probe.py

python3 scripts/pmc_summary.py stats "$(find $O/r02_prof_1080p -name "*_kernel_stats.csv" | tail -1)" $O/r02_1080p_dense8x8_kernel_stats.csv
python3 scripts/pmc_summary.py stats "$(find $O/r02_prof_fine -name "*_kernel_stats.csv" | tail -1)" $O/r02_4k_fine_shipped_env_kernel_stats.csv
python3 scripts/pmc_summary.py stats "$(find $O/r02_prof_merge -name "*_kernel_stats.csv" | tail -1)" $O/r02_merge_kernel_stats.csv
python3 scripts/pmc_summary.py stats "$(find $O/r02_prof_compact -name "*_kernel_stats.csv" | tail -1)" $O/r02_compact_probe_kernel_stats.csv
grep -E "aos40" $O/r02_prof_compact.log > $O/r02_compact_probe.txt
rm -f $O/r02_pmc_traffic.json
python3 scripts/pmc_summary.py pmc $O/r02_bench_1080p_dense8x8.json $O/r02_pmc_fetch_1080p $O/r02_pmc_write_1080p 1080p_dense8x8:4096 $O/r02_pmc_traffic.json
python3 scripts/pmc_summary.py pmc $O/r02_bench_4k_fine_shipped_env.json $O/r02_pmc_fetch_fine $O/r02_pmc_write_fine 4k_fine_shipped_env:1024 $O/r02_pmc_traffic.json
cat $O/r02_1080p_dense8x8_kernel_stats.csv $O/r02_4k_fine_shipped_env_kernel_stats.csv $O/r02_merge_kernel_stats.csv
