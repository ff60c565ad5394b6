#!/usr/bin/env python3
"""Round-4 probe: is the 64-stream host feed (145 k frames/s) below the single hot stream (190 k) because its 64 sources
(1 GB) come from DRAM?  The same 64 x 1 run with ONE 12-frame file given 64 times (one mapping, L3-resident) against 64
distinct files.  Needs a GPU."""
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
n, reps = 12, int(os.environ.get("REPS", "600"))
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    paths = []
    for k in range(64):
        spec = synth.spec_1080p(seed=2000 + k, sub=2)
        spec.events = [synth.Event(1, 1 + n // 2, 10 + k, 12 + k % 40, 4, 3, 9, 2)]
        frames = [synth.gen_frame(spec, i) for i in range(1, 1 + n)]
        path = os.path.join(d, f"cam{k:02d}.mtmv")
        m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, [spec.pts_ticks(i) for i in range(n)], frames, key=[1] * n)
        paths.append(path)
    env = dict(os.environ, CHUNK_DURATION_SEC="10", TARGET_FPS="0")
    def cpu_stat():
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
        except Exception:
            return {}
    runs = [("64 distinct files (1 GB of source, DRAM)", paths, None), ("one file 64 times (15.7 MB of source, L3)", [paths[0]] * 64, None)] * 2
    if os.environ.get("WINDOWS"):          # the same run confined to a window of CPUs (taskset): does the throttling go away?
        runs = []
        for w in os.environ["WINDOWS"].split(";"):
            runs += [("64 distinct files", paths, None if w == "none" else w)]
        runs = runs * 2
    for label, files, window in runs:
        pre = ["taskset", "-c", window] if window else []
        c0 = cpu_stat()
        r = subprocess.run(pre + [exe] + files + ["--streams", "64", "--threads", "1", "--repeat", str(reps), "--summary", "--outdir", d],
                           capture_output=True, text=True, env=env, timeout=300)
        c1 = cpu_stat()
        label = f"{label} [cpus {window or 'all'}] throttled {c1.get('nr_throttled', 0) - c0.get('nr_throttled', 0)}/{c1.get('nr_periods', 0) - c0.get('nr_periods', 0)} periods"
        s = [json.loads(ln)["batch_summary"] for ln in r.stdout.splitlines() if "batch_summary" in ln][0]
        busy = s["decode_us"] + s["analyze_us"]
        print(label, "steady", round(s["frames_scanned"] / (s["scan_window_us"] * 1e-6)), "wall", round(s["frames_scanned"] / (s["wall_us"] * 1e-6)),
              "copy", round(s["copy_us"] / busy, 2), "wait", round(s["wait_us"] / busy, 2), flush=True)
