#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (never the bench `value`):
  sync   mtgpu_scan_frames on pageable numpy memory (copy + scan + copy back, blocking)
  pipe   ScanPipe: pinned staging, 3 batches in flight, frames fed one by one
Prints frames/s and GB/s of MV bytes for 1080p dense8x8."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

spec = synth.spec_1080p(seed=3)
n = 120
spec.events = synth.scripted_events(spec, n)
frames = [synth.gen_frame(spec, i) for i in range(n)]
batch = m.FrameBatch.from_frames(frames)
nbytes = batch.mv.nbytes
s = m.MotionScanner(m.ScanParams.from_config(1920, 1080))
s.check_frames(batch)
t0 = time.perf_counter()
R = 5
for _ in range(R):
    fl = s.check_frames(batch)
dt = (time.perf_counter() - t0) / R
print(f"sync  pageable: {n / dt:9.0f} frames/s  {nbytes / dt / 1e9:6.2f} GB/s  ({dt * 1e3:.1f} ms per {n}-frame batch)")
pipe = m.ScanPipe(s, 32640 * 16, 16, 3)
for rep in range(2):
    t0 = time.perf_counter()
    for i, f in enumerate(frames):
        pipe.feed(f, spec.pts_seconds(i), i)
    out = pipe.drain()
    dt = time.perf_counter() - t0
assert [f for _, f, _ in out] == fl.tolist()
print(f"pipe  pinned x3: {n / dt:9.0f} frames/s  {nbytes / dt / 1e9:6.2f} GB/s  (python feed loop, 16-frame batches)")
