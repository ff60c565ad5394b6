#!/usr/bin/env python3
"""Frames per workgroup (MTGPU_GROUP, read at mtgpu_create) against the automatic choice, interleaved in one process:
wall clock per call over back-to-back calls.  Usage: group_ab.py workload frames [compact] — groups from GROUPS (1,2,4)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "1080p_dense8x8"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
compact = len(sys.argv) > 3 and sys.argv[3] == "compact"
dev = torch.device("cuda", 0)
w = bench.build_workload(wl, "code_defaults", frames, 60, 1000, dev)
builds = [("auto", w["scanner"])]
for g in os.environ.get("GROUPS", "1,2,4").split(","):      # (a trailing "p", as in the logs of round 6, once forced the work list)
    os.environ["MTGPU_GROUP"] = g.rstrip("p")
    builds.append((f"group{g}", m.MotionScanner(w["params"], 0)))
    os.environ.pop("MTGPU_GROUP")
if compact:
    rec = m.pack_records(w["mv"])
    d_in = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev).repeat(w["reps"])[: w["n_records"] * 8].contiguous()
    nbytes = 8 * w["n_records"] + 9 * frames
else:
    d_in = w["d_mv"]
    nbytes = w["alg_bytes"]
N = int(os.environ.get("CALLS", "50"))
res = {name: [] for name, _ in builds}
ref = None
for rnd in range(5):
    for name, s in builds:
        fn = s.check_frames_device_compact if compact else s.check_frames_device
        for _ in range(3):
            fn(d_in, w["d_off"], None, w["d_flags"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            fn(d_in, w["d_off"], None, w["d_flags"])
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / N * 1e6)
        fl = w["d_flags"].cpu().numpy()
        ref = fl if ref is None else ref
        assert np.array_equal(fl, ref), name
for name, _ in builds:
    t = float(np.median(res[name]))
    print(f"{wl} {frames} {'compact' if compact else 'aos40'} {name:8s} {t:9.2f} us per call  {nbytes / t / 1e3:7.0f} GB/s", flush=True)
