#!/bin/bash
# Round 4: SQ / LDS counters of the banded 960x540 scan kernel on typical input, on vote-heavy input with one record per
# block (0.65-0.71 of peak) and with four (0.73): what does the vote path cost where it still costs?  Separate --pmc passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_pmc_votes
mkdir -p $O
PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"
        "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
        "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
        "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_FLAT")
run_set() {   # name pan workload
  name=$1; pan=$2; wl=$3; i=0
  for p in "${PASSES[@]}"; do
    d=$O/${name}_$i
    AB_PAN=$pan timeout -k 10 150 rocprofv3 --kernel-trace --pmc $p -f csv -d $d -- python3 bench.py --workload $wl --params shipped_env --frames 512 --steps 3 --warmup 1 --cpu-seconds 0 --no-others --no-host --no-merge > $d.log 2>&1 || { echo "pass $name $i ($p) failed"; tail -2 $d.log; }
    i=$((i+1))
  done
}
run_set typical_dense4 0 4k_fine_dense4
run_set pan_one_record 1 4k_fine
run_set pan_dense4 1 4k_fine_dense4
python3 - <<'PY'
import csv, glob, collections, json, os
O = "gpurun_out/r04_pmc_votes"
res = collections.defaultdict(dict)
for f in glob.glob(O + "/*/**/*_counter_collection.csv", recursive=True):
    w = os.path.relpath(f, O).split("/")[0].rsplit("_", 1)[0]
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f, newline="")):
        if "scan_frames_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        res[w][k] = sum(v) / len(v)
json.dump(res, open("gpurun_out/r04_pmc_votes.json", "w"), indent=1)
for w, d in res.items():
    print(w, {k: round(v) for k, v in sorted(d.items())})
PY
rm -rf $O
