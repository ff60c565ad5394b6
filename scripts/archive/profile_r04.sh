#!/bin/bash
# Round-4 profile collection on the GPU box (run from the repo root).
#  (1) VERDICT r3 item 4: the vote-heavy ("camera pan", AB_PAN=1: every record above the threshold) case of the
#      banded 960x540 plan — one `rocprofv3 --kernel-trace --stats` pass and, in SEPARATE passes, FETCH_SIZE and
#      WRITE_SIZE (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r04_pmc_traffic.json key "...:pan".
#  (2) the headline leg again on this round's kernels (stats only; the kernel changed by one SGPR compare).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
COMMON="--cpu-seconds 0 --no-others --no-host"
run() { name=$1; shift; echo "== $name $(date +%T)"; "$@" > $O/$name.log 2>&1 || { tail -5 $O/$name.log; exit 1; }; }
rm -f $O/r04_pmc_traffic.json
leg() {   # tag workload params frames steps key
  tag=$1; wl=$2; pn=$3; fr=$4; st=$5; key=$6
  A="--workload $wl --params $pn --frames $fr $COMMON"
  run ${tag}_stats rocprofv3 --kernel-trace --stats -f csv -d $O/${tag}_stats -- python3 bench.py $A --steps $st --warmup 3
  grep '^{' $O/${tag}_stats.log | tail -1 > $O/${tag}_bench.json
  python3 scripts/pmc_summary.py stats "$(find $O/${tag}_stats -name "*_kernel_stats.csv" | tail -1)" $O/${tag}_kernel_stats.csv
  if [ -n "$key" ]; then
    run ${tag}_fetch rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/${tag}_fetch -- python3 bench.py $A --steps 3 --warmup 1
    run ${tag}_write rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/${tag}_write -- python3 bench.py $A --steps 3 --warmup 1
    python3 scripts/pmc_summary.py pmc $O/${tag}_bench.json $O/${tag}_fetch $O/${tag}_write $key $O/r04_pmc_traffic.json "round 4, scripts/profile_r04.sh"
  fi
  rm -rf $O/${tag}_stats $O/${tag}_fetch $O/${tag}_write
}
export AB_PAN=1
leg r04_4k_fine_dense4_shipped_env_pan 4k_fine_dense4 shipped_env 1024 8 4k_fine_dense4:shipped_env:1024:pan
leg r04_4k_fine_code_defaults_pan 4k_fine code_defaults 1024 8 4k_fine:code_defaults:1024:pan
unset AB_PAN
leg r04_4k_fine_dense4_shipped_env 4k_fine_dense4 shipped_env 1024 8 4k_fine_dense4:shipped_env:1024
leg r04_1080p_dense8x8_code_defaults_16384 1080p_dense8x8 code_defaults 16384 20 ""
cat $O/r04_*_kernel_stats.csv | grep -E "scan_frames|Name" | cut -c1-200
cat $O/r04_pmc_traffic.json
