#!/bin/bash
# Generalised from scripts/profile_r03.sh: R=r04 (default) names the round's output files.
R=${R:-r04}
# Per-round profile collection on the GPU box (run from the repo root): for EVERY leg of the bench line
# (headline + other_workloads) one `rocprofv3 --kernel-trace --stats` pass and, in SEPARATE passes, the
# FETCH_SIZE and WRITE_SIZE counters (MI355X_MICROARCH.md, HBM section: never together, never with a
# trace domain other than --kernel-trace).  The program stands directly after `--`.
# Summaries land in gpurun_out/${R}_*; scripts/pmc_summary.py merges the PMC rows into
# gpurun_out/${R}_pmc_traffic.json (copied to profiles/pmc_traffic.json, which bench.py replays).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
COMMON="--cpu-seconds 0 --no-others --no-host"
run() { name=$1; shift; echo "== $name $(date +%T)"; "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }; }
rm -f $O/${R}_pmc_traffic.json
# workload params frames steps
LEGS="1080p_dense8x8:code_defaults:16384:20 1080p_dense8x8:shipped_env:16384:20 1080p_dense8x8:code_defaults:4096:20 4k_dense8x8:code_defaults:4096:20 4k_dense8x8:code_defaults:1024:20 4k_fine:code_defaults:1024:8 \
4k_fine_dense4:shipped_env:1024:8 1080p_dense8x8:shipped_env:4096:20 1080p_dense16:code_defaults:65536:20 1080p_dense16:code_defaults:16384:20 480p_dense16:code_defaults:262144:20"
for leg in ${LEGS_OVERRIDE:-$LEGS}; do
  IFS=: read wl pn fr st <<< "$leg"
  tag=${R}_${wl}_${pn}_${fr}
  A="--workload $wl --params $pn --frames $fr $COMMON"
  run ${tag}_stats rocprofv3 --kernel-trace --stats -f csv -d $O/${tag}_stats -- python3 bench.py $A --steps $st --warmup 3
  grep '^{' $O/${tag}_stats.log | tail -1 > $O/${tag}_bench.json
  python3 scripts/pmc_summary.py stats "$(find $O/${tag}_stats -name "*_kernel_stats.csv" | tail -1)" $O/${tag}_kernel_stats.csv
  run ${tag}_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/${tag}_fetch -- python3 bench.py $A --steps 3 --warmup 1
  run ${tag}_write rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/${tag}_write -- python3 bench.py $A --steps 3 --warmup 1
  python3 scripts/pmc_summary.py pmc $O/${tag}_bench.json $O/${tag}_fetch $O/${tag}_write $wl:$pn:$fr $O/${R}_pmc_traffic.json "round ${R#r0}, scripts/profile_legs.sh"
  rm -rf $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write
done
if [ -z "$LEGS_OVERRIDE" ]; then
  run ${R}_prof_compact rocprofv3 --kernel-trace --stats -f csv -d $O/${R}_prof_compact -- python3 scripts/compact_probe.py
  python3 scripts/pmc_summary.py stats "$(find $O/${R}_prof_compact -name "*_kernel_stats.csv" | tail -1)" $O/${R}_compact_probe_kernel_stats.csv
  grep -E "aos40" $O/${R}_prof_compact.log > $O/${R}_compact_probe.txt
  rm -rf $O/${R}_prof_compact
fi
cat $O/${R}_*_kernel_stats.csv | grep -E "scan_frames|Name" | cut -c1-200
