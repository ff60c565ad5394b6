#!/bin/bash
# Round 4: the GPU's view of the host feed.  rocprofv3 --kernel-trace around mtgpu_scan_file for (a) one hot stream with
# 16 workers and (b) 64 streams x 1 worker: per-kernel duration, how much of the time a scan kernel was running, and the
# PCIe rate while one was.  (The program stands directly after `--`.)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_kview
mkdir -p $O /dev/shm/kview
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import mvtrim_amd as m
from mvtrim_amd import synth
n = 12
for k in range(64):
    spec = synth.spec_1080p(seed=2000 + k, sub=2)
    spec.events = [synth.Event(1, 1 + n // 2, 10 + k, 12 + k % 40, 4, 3, 9, 2)]
    frames = [synth.gen_frame(spec, i) for i in range(1, 1 + n)]
    m.mvfile.write_mtmv(f"/dev/shm/kview/cam{k:02d}.mtmv", 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps,
                        [spec.pts_ticks(i) for i in range(n)], frames, key=[1] * n)
PY
export CHUNK_DURATION_SEC=10 TARGET_FPS=0
EXE=motion-estimated-video-trimmer_amd/mtgpu_scan_file
rocprofv3 --kernel-trace -f csv -d $O/hot -- $EXE /dev/shm/kview/cam00.mtmv --threads 16 --repeat 5000 > $O/hot.json 2> $O/hot.err || { tail -3 $O/hot.err; exit 1; }
rocprofv3 --kernel-trace -f csv -d $O/s64 -- $EXE /dev/shm/kview/cam*.mtmv --streams 64 --threads 1 --repeat 300 --summary --outdir /dev/shm/kview > $O/s64.json 2> $O/s64.err || { tail -3 $O/s64.err; exit 1; }
python3 - <<'PY'
import csv, glob, json
for name, frames in (("hot", 12 * 5000), ("s64", 64 * 12 * 300)):
    tr = glob.glob(f"gpurun_out/r04_kview/{name}/**/*kernel_trace.csv", recursive=True)[-1]
    t = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr)) if "scan_frames" in r["Kernel_Name"])
    dur = [e - s for s, e in t]
    busy = 0; cs, ce = t[0]; conc = 0
    for s, e in t[1:]:
        if s > ce: busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    span = t[-1][1] - t[0][0]
    gb = frames * 32640 * 8 / 1e9
    dur.sort()
    print(name, "kernels", len(t), "median us", round(dur[len(dur)//2] / 1e3, 1), "p90 us", round(dur[int(len(dur)*0.9)] / 1e3, 1),
          "sum of durations / span", round(sum(dur) / span, 2), "(average kernels in flight)",
          "| a scan kernel running", round(busy / span, 3), "of the span | frames/s over span", round(frames / (span * 1e-9)),
          "| PCIe GB/s over span", round(gb / (span * 1e-9), 1), "while busy", round(gb / (busy * 1e-9), 1))
PY
rm -rf $O/hot $O/s64 /dev/shm/kview
