#!/bin/bash
# Per-round profile collection on the GPU box (run from the repo root; R=r06 names the output files): for EVERY leg of
# the bench line (headline + other_workloads) and the vote-heavy ("pan") cases of the fine grid, one
# `rocprofv3 --kernel-trace --stats` pass and, in SEPARATE passes, the FETCH_SIZE and WRITE_SIZE counters
# (MI355X_MICROARCH.md, HBM section: never together, never with a trace domain other than --kernel-trace).  The program
# stands directly after `--`.  Summaries land in gpurun_out/${R}_*; scripts/pmc_summary.py merges the PMC rows into
# gpurun_out/${R}_pmc_traffic.json (copied to profiles/pmc_traffic.json, which bench.py replays for the legs it does
# not measure itself).
R=${R:-r06}
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
COMMON="--cpu-seconds 0 --no-others --no-host --no-pmc"
run() { name=$1; shift; echo "== $name $(date +%T)"; "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }; }
[ -n "$KEEP_PMC" ] || rm -f $O/${R}_pmc_traffic.json
# workload:params:frames:steps[:pan]
LEGS="1080p_dense8x8:code_defaults:16384:20 1080p_dense8x8:shipped_env:16384:20 4k_dense8x8:code_defaults:4096:20 4k_dense8x8:shipped_env:4096:20 \
4k_fine:code_defaults:1024:8 4k_fine_dense4:shipped_env:1024:8 1080p_dense16:code_defaults:65536:20 480p_dense16:code_defaults:262144:20 \
1080p_dense8x8:code_defaults:4096:20 4k_dense8x8:code_defaults:1024:20 \
4k_fine_dense4:shipped_env:1024:8:pan 4k_fine:shipped_env:1024:8:pan 4k_fine:code_defaults:1024:8:pan"
for leg in ${LEGS_OVERRIDE:-$LEGS}; do
  IFS=: read wl pn fr st pan <<< "$leg"
  tag=${R}_${wl}_${pn}_${fr}${pan:+_pan}
  key=$wl:$pn:$fr${pan:+:pan}
  if [ -n "$pan" ]; then export AB_PAN=1; else unset AB_PAN; fi
  A="--workload $wl --params $pn --frames $fr $COMMON"
  run ${tag}_stats rocprofv3 --kernel-trace --stats -f csv -d $O/${tag}_stats -- python3 bench.py $A --steps $st --warmup 3
  grep '^{' $O/${tag}_stats.log | tail -1 > $O/${tag}_bench.json
  python3 scripts/pmc_summary.py stats "$(find $O/${tag}_stats -name "*_kernel_stats.csv" | tail -1)" $O/${tag}_kernel_stats.csv
  run ${tag}_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/${tag}_fetch -- python3 bench.py $A --steps 3 --warmup 1
  run ${tag}_write rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/${tag}_write -- python3 bench.py $A --steps 3 --warmup 1
  python3 scripts/pmc_summary.py pmc $O/${tag}_bench.json $O/${tag}_fetch $O/${tag}_write $key $O/${R}_pmc_traffic.json "round ${R#r0}, scripts/profile_round.sh"
  rm -rf $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write $O/${tag}_fetch.log $O/${tag}_write.log
done
unset AB_PAN
cat $O/${R}_*_kernel_stats.csv | grep -E "scan_frames|plan_|Name" | cut -c1-200
