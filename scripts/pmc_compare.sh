#!/bin/bash
# Memory-path counters of the scan kernel for two workloads side by side (separate --pmc passes):
# why does a 960x540 frame (20.7 MB per workgroup) stream slower than a 1080p frame (1.3 MB)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02_pmc_cmp
mkdir -p $O
declare -A WL
WL[1080p]="--workload 1080p_dense8x8 --frames 4096"
WL[fine]="--workload 4k_fine --frames 1024"
WL[4k]="--workload 4k_dense8x8 --frames 1024"
PASSES=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
        "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
        "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
        "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"
        "GRBM_GUI_ACTIVE TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum")
for w in 1080p fine 4k; do
  i=0
  for p in "${PASSES[@]}"; do
    d=$O/${w}_$i
    timeout -k 10 120 rocprofv3 --kernel-trace --pmc $p -f csv -d $d -- python3 bench.py ${WL[$w]} --steps 3 --warmup 1 --cpu-seconds 0 --no-others --no-host --no-merge > $d.log 2>&1 || { echo "pass $w $i failed"; tail -3 $d.log; }
    i=$((i+1))
  done
done
python3 - <<'PY'
import csv, glob, collections, json, os
O = "gpurun_out/r02_pmc_cmp"
res = collections.defaultdict(dict)
for f in glob.glob(O + "/*/**/*_counter_collection.csv", recursive=True):
    w = os.path.relpath(f, O).split("/")[0].rsplit("_", 1)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f, newline="")):
        if "scan_frames_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[w][k] = sum(v) / len(v)
json.dump(res, open("gpurun_out/r02_pmc_compare.json", "w"), indent=1)
for w, d in res.items():
    print(w, {k: round(v, 1) for k, v in sorted(d.items())})
PY
