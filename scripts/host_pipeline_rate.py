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
distinct, n = 150, 3000                                   # 100 s at 30 fps, 3.9 GB of records
spec.events = synth.scripted_events(spec, distinct)
tile = [synth.gen_frame(spec, i) for i in range(distinct)]
frames = [tile[i % distinct] for i in range(n)]
ticks = [spec.pts_ticks(i) for i in range(n)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    path = os.path.join(d, "s.mtmv")
    m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
    size = os.path.getsize(path)
    env = dict(os.environ, CHUNK_DURATION_SEC="5", TARGET_FPS="0")
    for threads in (1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        out = subprocess.run([exe, path, "--threads", str(threads)], check=True, capture_output=True, text=True, env=env).stdout
        dt = time.perf_counter() - t0
        r = json.loads(out)
        sw = r["scan_wall_us"] / 1e6
        work = max(sw - r["init_us"] / 1e6 / threads, 1e-9)
        print(f"threads={threads:2d}  process {dt:5.2f} s | scan phase {sw:5.2f} s (worker init {r['init_us'] / 1e6 / threads:.2f} s each) "
              f"-> {n / work:8.0f} frames/s  {size / work / 1e9:6.2f} GB/s after init  "
              f"(analyze {r['analyze_us'] / 1e6:.2f} s summed; {r['motion_frames']} motion frames)")
