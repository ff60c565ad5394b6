#!/usr/bin/env python3
"""Same 1080p batch, code defaults vs shipped env (and mixtures), interleaved in one process: which parameter
costs the shipped-env leg its 2-5 %?  Usage: python scripts/ab_params.py [frames]   (needs a GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402
import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
w = bench.build_workload("1080p_dense8x8", "code_defaults", frames, 60, 1000, dev)
variants = [("T16 vec2 (code defaults)", dict(mv_threshold_sq=16.0, vectors_needed=2)),
            ("T4 vec4 (shipped env)", dict(mv_threshold_sq=4.0, vectors_needed=4)),
            ("T4 vec2", dict(mv_threshold_sq=4.0, vectors_needed=2)),
            ("T16 vec4", dict(mv_threshold_sq=16.0, vectors_needed=4)),
            ("T1 vec2 (a third of the background votes)", dict(mv_threshold_sq=1.0, vectors_needed=2)),
            ("T0 vec2 (every record votes)", dict(mv_threshold_sq=0.0, vectors_needed=2))]
scanners = []
for name, kw in variants:
    p = m.ScanParams.from_config(1920, 1080, **kw)
    scanners.append((name, m.MotionScanner(p, 0), torch.empty(frames, dtype=torch.uint8, device=dev), []))
for r in range(14):
    for name, s, fl, times in scanners:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        for e0, e1 in evs:
            e0.record()
            s.check_frames_device(w["d_mv"], w["d_off"], None, fl)
            e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            times.extend(e0.elapsed_time(e1) for e0, e1 in evs[1:])
for name, s, fl, times in scanners:
    t = np.array(times)
    print(f"{name:46s} median {np.median(t):.4f} ms  min {t.min():.4f}  {w['alg_bytes'] / np.median(t) / 1e6:7.0f} GB/s  motion {int(fl.sum())}")
