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
distinct, n = 150, 12000                                  # 400 s at 30 fps, 15.7 GB of records
spec.events = synth.scripted_events(spec, distinct)
tile = [synth.gen_frame(spec, i) for i in range(distinct)]
frames = [tile[i % distinct] for i in range(n)]
ticks = [spec.pts_ticks(i) for i in range(n)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    path = os.path.join(d, "s.mtmv")
    m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
    size = os.path.getsize(path)
    rows = []
    sweep = [("aos40", 16), ("compact8", 4), ("compact8_zc", 4), ("aos40_zc", 16)]
    if os.environ.get("RATE_SWEEP") == "1":
        sweep = [("aos40", 16), ("aos40", 4), ("compact8", 1), ("compact8", 2), ("compact8", 4), ("compact8", 8), ("compact8", 16)]
    for staging, batch_mb in sweep:
        env = dict(os.environ, CHUNK_DURATION_SEC="10", TARGET_FPS="0", MTGPU_STAGING=staging, MTGPU_BATCH_MB=str(batch_mb))
        for threads in ((1, 4, 8, 16, 32) if os.environ.get("RATE_SWEEP") == "1" else (1, 2, 4, 8, 16)):
            best = None
            for rep in range(2):                               # second pass: page cache and clocks warm
                t0 = time.perf_counter()
                out = subprocess.run([exe, path, "--threads", str(threads)], check=True, capture_output=True, text=True, env=env).stdout
                dt = time.perf_counter() - t0
                r = json.loads(out)
                sw = r["scan_wall_us"] / 1e6
                work = max(r["scan_work_us"] / 1e6, 1e-9)       # from "all workers initialised" to the last result
                row = {"staging": staging, "batch_mib": batch_mb, "threads": threads, "process_s": dt, "scan_phase_s": sw,
                       "worker_init_s": r["init_us"] / 1e6 / threads, "frames_per_s": n / work,
                       "aos_GBps": size / work / 1e9, "analyze_s_summed": r["analyze_us"] / 1e6,
                       "decode_s_summed": r["decode_us"] / 1e6, "copy_s_summed": r["copy_us"] / 1e6,
                       "submit_s_summed": r["submit_us"] / 1e6, "gpu_wait_s_summed": r["wait_us"] / 1e6,
                       "motion_frames": r["motion_frames"]}
                if best is None or row["frames_per_s"] > best["frames_per_s"]:
                    best = row
            rows.append(best)
            print(f"{staging:11s} batch={batch_mb:2d}MiB threads={threads:2d}  process {best['process_s']:5.2f} s | scan phase {best['scan_phase_s']:5.2f} s "
                  f"(worker init {best['worker_init_s']:.2f} s each) -> {best['frames_per_s']:8.0f} frames/s  "
                  f"{best['aos_GBps']:6.2f} GB/s of AVMotionVector bytes after init  (analyze {best['analyze_s_summed']:.2f} s summed = copy "
                  f"{best['copy_s_summed']:.2f} + submit {best['submit_s_summed']:.2f} + wait {best['gpu_wait_s_summed']:.2f}; "
                  f"{best['motion_frames']} motion frames)", flush=True)
    # ---- hot source: a 12-frame stream (16 MB: stays in the workers' caches, like side data a decoder
    # thread has just written) repeated 1000 times = 12000 frames
    hot_rows = []
    hot_n = 12
    hpath = os.path.join(d, "hot.mtmv")
    hframes = [tile[1 + i] for i in range(hot_n)]
    m.mvfile.write_mtmv(hpath, 1920, 1080, 1, spec.tb_den, spec.fps, hot_n / spec.fps,
                        [spec.pts_ticks(i) for i in range(hot_n)], hframes, key=[1] * hot_n)
    reps = 1000
    for staging, batch_mb in (("aos40", 16), ("compact8", 4), ("compact8_zc", 4), ("compact8_zc", 16), ("aos40_zc", 16)):
        env = dict(os.environ, CHUNK_DURATION_SEC="10", TARGET_FPS="0", MTGPU_STAGING=staging, MTGPU_BATCH_MB=str(batch_mb))
        for threads in (1, 4, 16):
            best = None
            for rep in range(2):
                out = subprocess.run([exe, hpath, "--threads", str(threads), "--repeat", str(reps)], check=True,
                                     capture_output=True, text=True, env=env).stdout
                r = json.loads(out)
                work = max(r["scan_work_us"] / 1e6, 1e-9)
                nfr = hot_n * reps
                row = {"staging": staging, "batch_mib": batch_mb, "threads": threads, "frames_per_s": nfr / work,
                       "aos_GBps": nfr * 32640 * 40 / work / 1e9, "copy_s_summed": r["copy_us"] / 1e6,
                       "submit_s_summed": r["submit_us"] / 1e6, "gpu_wait_s_summed": r["wait_us"] / 1e6}
                if best is None or row["frames_per_s"] > best["frames_per_s"]:
                    best = row
            hot_rows.append(best)
            print(f"hot source {staging:11s} {batch_mb:2d}MiB threads={threads:2d} -> {best['frames_per_s']:8.0f} frames/s  {best['aos_GBps']:6.2f} GB/s of "
                  f"AVMotionVector bytes (copy {best['copy_s_summed']:.2f} + submit {best['submit_s_summed']:.2f} + wait "
                  f"{best['gpu_wait_s_summed']:.2f} s summed)", flush=True)
    outp = os.path.join(ROOT, "gpurun_out", "r02_host_pipeline_rate.json")
    if os.path.isdir(os.path.dirname(outp)):
        json.dump({"what": "mtgpu_scan_file on a 12000-frame 1080p dense8x8 .mtmv in /dev/shm (15.7 GB of records): mmap -> "
                           "pinned staging -> H2D -> scan -> merge; PCIe-inclusive, never the bench value",
                   "frames": n, "file_bytes": size, "rows": rows,
                   "hot_source": {"what": "the same pipeline fed from a 12-frame (16 MB) stream repeated 1000x: MV bytes "
                                          "cache-resident, as when a decoder thread has just written them", "rows": hot_rows}}, open(outp, "w"), indent=1)
