#!/usr/bin/env python3
"""Batch mode of the C++ host pipeline (process_batch): N videos x S streams x T workers through
mtgpu_scan_file on 1080p dense8x8 .mtmv files.  PCIe-inclusive; stability + rate check."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

exe = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "mtgpu_scan_file")
spec = synth.spec_1080p(seed=9)
distinct, n = 120, 1200                                   # 40 s at 30 fps, 1.5 GB of records per file
spec.events = synth.scripted_events(spec, distinct)
tile = [synth.gen_frame(spec, i) for i in range(distinct)]
frames = [tile[i % distinct] for i in range(n)]
ticks = [spec.pts_ticks(i) for i in range(n)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    first = os.path.join(d, "v0.mtmv")
    m.mvfile.write_mtmv(first, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
    size = os.path.getsize(first)
    files = [first]
    for k in range(1, 8):
        p = os.path.join(d, f"v{k}.mtmv")
        shutil.copy(first, p)
        files.append(p)
    env = dict(os.environ, CHUNK_DURATION_SEC="5", TARGET_FPS="0")
    for streams, threads in ((1, 4), (4, 2), (8, 2), (8, 4)):
        t0 = time.perf_counter()
        out = subprocess.run([exe] + files + ["--streams", str(streams), "--threads", str(threads), "--outdir", d],
                             check=True, capture_output=True, text=True, env=env).stdout
        dt = time.perf_counter() - t0
        jobs = [json.loads(ln) for ln in out.strip().splitlines()]
        assert len(jobs) == 8 and len({json.dumps(j["segments"]) for j in jobs}) == 1
        print(f"8 videos, streams={streams} x threads={threads}: process wall {dt:5.2f} s -> {8 * n / dt:8.0f} frames/s "
              f"{8 * size / dt / 1e9:6.2f} GB/s (incl. start-up)")
