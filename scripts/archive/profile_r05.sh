#!/bin/bash
# Round-5 profile collection on the GPU box (run from the repo root).
#  (1) VERDICT r4 item 5 (r3 item 4 before it): the vote-heavy ("camera pan", AB_PAN=1: every record above the threshold) case of the
#      banded 960x540 plan — one `rocprofv3 --kernel-trace --stats` pass and, in SEPARATE passes, FETCH_SIZE and
#      WRITE_SIZE (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r05_pmc_traffic.json key "...:pan".
#  (2) the same plan on typical input.  Every other bench leg: R=r05 scripts/profile_legs.sh.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
COMMON="--cpu-seconds 0 --no-others --no-host"
run() { name=$1; shift; echo "== $name $(date +%T)"; "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }; }
[ -n "$KEEP_PMC" ] || rm -f $O/r05_pmc_traffic.json
leg() {   # tag workload params frames steps key
  tag=$1; wl=$2; pn=$3; fr=$4; st=$5; key=$6
  A="--workload $wl --params $pn --frames $fr $COMMON"
  run ${tag}_stats rocprofv3 --kernel-trace --stats -f csv -d $O/${tag}_stats -- python3 bench.py $A --steps $st --warmup 3
  grep '^{' $O/${tag}_stats.log | tail -1 > $O/${tag}_bench.json
  python3 scripts/pmc_summary.py stats "$(find $O/${tag}_stats -name "*_kernel_stats.csv" | tail -1)" $O/${tag}_kernel_stats.csv
  if [ -n "$key" ]; then
    run ${tag}_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/${tag}_fetch -- python3 bench.py $A --steps 3 --warmup 1
    run ${tag}_write rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/${tag}_write -- python3 bench.py $A --steps 3 --warmup 1
    python3 scripts/pmc_summary.py pmc $O/${tag}_bench.json $O/${tag}_fetch $O/${tag}_write $key $O/r05_pmc_traffic.json "round 5, scripts/profile_r05.sh"
  fi
  rm -rf $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write
}
export AB_PAN=1
leg r05_4k_fine_dense4_shipped_env_pan 4k_fine_dense4 shipped_env 1024 8 4k_fine_dense4:shipped_env:1024:pan
leg r05_4k_fine_shipped_env_pan 4k_fine shipped_env 1024 8 4k_fine:shipped_env:1024:pan
leg r05_4k_fine_code_defaults_pan 4k_fine code_defaults 1024 8 4k_fine:code_defaults:1024:pan
unset AB_PAN
leg r05_4k_fine_dense4_shipped_env 4k_fine_dense4 shipped_env 1024 8 4k_fine_dense4:shipped_env:1024
cat $O/r05_*_kernel_stats.csv | grep -E "scan_frames|Name" | cut -c1-200
cat $O/r05_pmc_traffic.json
