#!/usr/bin/env python3
"""End-to-end rate of the C++ host pipeline (mtgpu_scan_file: mmap'ed .mtmv -> pinned staging ->
H2D -> scan -> merge) on a 1080p dense8x8 stream, for several worker counts.  PCIe-inclusive;
never the bench value.  Needs a GPU."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

exe = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "mtgpu_scan_file")
spec = synth.spec_1080p(seed=9)
distinct, n = 150, 6000                                   # 200 s at 30 fps, 7.8 GB of records
spec.events = synth.scripted_events(spec, distinct)
tile = [synth.gen_frame(spec, i) for i in range(distinct)]
frames = [tile[i % distinct] for i in range(n)]
ticks = [spec.pts_ticks(i) for i in range(n)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    path = os.path.join(d, "s.mtmv")
    m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
    size = os.path.getsize(path)
    rows = []
    for staging in ("aos40", "compact8"):
        env = dict(os.environ, CHUNK_DURATION_SEC="5", TARGET_FPS="0", MTGPU_STAGING=staging)
        for threads in (1, 2, 4, 8, 16):
            best = None
            for rep in range(2):                               # second pass: page cache and clocks warm
                t0 = time.perf_counter()
                out = subprocess.run([exe, path, "--threads", str(threads)], check=True, capture_output=True, text=True, env=env).stdout
                dt = time.perf_counter() - t0
                r = json.loads(out)
                sw = r["scan_wall_us"] / 1e6
                work = max(r["scan_work_us"] / 1e6, 1e-9)       # from "all workers initialised" to the last result
                row = {"staging": staging, "threads": threads, "process_s": dt, "scan_phase_s": sw,
                       "worker_init_s": r["init_us"] / 1e6 / threads, "frames_per_s": n / work,
                       "aos_GBps": size / work / 1e9, "analyze_s_summed": r["analyze_us"] / 1e6,
                       "decode_s_summed": r["decode_us"] / 1e6, "motion_frames": r["motion_frames"]}
                if best is None or row["frames_per_s"] > best["frames_per_s"]:
                    best = row
            rows.append(best)
            print(f"{staging:8s} threads={threads:2d}  process {best['process_s']:5.2f} s | scan phase {best['scan_phase_s']:5.2f} s "
                  f"(worker init {best['worker_init_s']:.2f} s each) -> {best['frames_per_s']:8.0f} frames/s  "
                  f"{best['aos_GBps']:6.2f} GB/s of AVMotionVector bytes after init  (analyze {best['analyze_s_summed']:.2f} s summed; "
                  f"{best['motion_frames']} motion frames)", flush=True)
    outp = os.path.join(ROOT, "gpurun_out", "r02_host_pipeline_rate.json")
    if os.path.isdir(os.path.dirname(outp)):
        json.dump({"what": "mtgpu_scan_file on a 6000-frame 1080p dense8x8 .mtmv in /dev/shm (7.8 GB of records): mmap -> "
                           "pinned staging -> H2D -> scan -> merge; PCIe-inclusive, never the bench value",
                   "frames": n, "file_bytes": size, "rows": rows}, open(outp, "w"), indent=1)
