#!/usr/bin/env python3
"""Where the host-fed pipeline's worker time goes (developer tool, needs a GPU): mtgpu_scan_file on a
12-frame 1080p dense8x8 stream repeated 500x (cache-resident source), default staging, for several
worker counts; prints frames/s and each worker's mean copy-out / submit / wait share."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

exe = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "mtgpu_scan_file")
spec = synth.spec_1080p(seed=9)
n, reps = 12, 500
spec.events = synth.scripted_events(spec, 60)
frames = [synth.gen_frame(spec, i) for i in range(31, 31 + n)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    path = os.path.join(d, "hot.mtmv")
    m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, [spec.pts_ticks(i) for i in range(n)],
                        frames, key=[1] * n)
    for staging in (os.environ.get("STAGINGS", "compact8_zc,compact8").split(",")):
        for threads in [int(x) for x in os.environ.get("THREADS", "4,8,16,32").split(",")]:
            env = dict(os.environ, CHUNK_DURATION_SEC="10", TARGET_FPS="0", MTGPU_STAGING=staging)
            best = None
            for _ in range(2):
                j = json.loads(subprocess.run([exe, path, "--threads", str(threads), "--repeat", str(reps)], check=True,
                                              capture_output=True, text=True, env=env).stdout)
                if best is None or j["scan_work_us"] < best["scan_work_us"]:
                    best = j
            w = best["scan_work_us"]
            print(f"{staging:12s} threads={threads:2d}  {n * reps / (w * 1e-6):9.0f} frames/s  per worker: "
                  f"copy {best['copy_us'] / threads / w:5.1%}  submit {best['submit_us'] / threads / w:5.1%}  "
                  f"wait {best['wait_us'] / threads / w:5.1%}  decode(source) {best['decode_us'] / threads / w:5.1%}", flush=True)
