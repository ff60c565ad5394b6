#!/usr/bin/env python3
"""Probe: scan kernel reading the record array directly from pinned HOST memory over PCIe
(no H2D copy, no device buffer) vs the staged path.  Needs a GPU."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

spec = synth.spec_1080p(seed=3)
n = 240
spec.events = synth.scripted_events(spec, 60)
tile = [synth.gen_frame(spec, i) for i in range(60)]
frames = [tile[i % 60] for i in range(n)]
b = m.FrameBatch.from_frames(frames)
s = m.MotionScanner(m.ScanParams.from_config(1920, 1080))
want = s.check_frames(b)
host = torch.from_numpy(b.mv.view(np.uint8).copy()).pin_memory()
d_off = torch.from_numpy(b.frame_off.astype(np.int64)).cuda()
fl = torch.empty(n, dtype=torch.uint8, device="cuda")
for label, tensor in (("pinned host (zero-copy)", host), ("device (resident)", host.cuda())):
    for _ in range(2):
        s.check_frames_device(tensor, d_off, None, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 5
    for _ in range(R):
        s.check_frames_device(tensor, d_off, None, fl)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    ok = np.array_equal(fl.cpu().numpy(), want)
    print(f"{label:26s} {dt * 1e3:8.3f} ms per {n} frames  {host.numel() / dt / 1e9:8.2f} GB/s  parity={ok}")
t0 = time.perf_counter()
for _ in range(5):
    dd = host.cuda(non_blocking=True)
    s.check_frames_device(dd, d_off, None, fl)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"{'H2D copy + scan (staged)':26s} {dt * 1e3:8.3f} ms per {n} frames  {host.numel() / dt / 1e9:8.2f} GB/s")
