#!/usr/bin/env python3
"""A/B of the current libmtgpu.so against a previous build (scripts/libmtgpu_prev.so, built by
hand from an older commit) in ONE process, interleaved.  Usage: ab_prev.py workload frames"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402
from bench import make_spec  # noqa: E402
from mvtrim_amd import synth, scanner as sc  # noqa: E402

wl, frames = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
spec, (W, H, gridkw) = make_spec(wl, seed=1)
distinct = 30
spec.events = synth.scripted_events(spec, distinct)
mv, off, pts, sd = synth.gen_stream(spec, distinct)
kw = dict(m.config.CODE_DEFAULTS)
kw.update(gridkw)
if spec.sub == 1:
    kw["vectors_needed"] = 1
if os.environ.get("AB_VEC"):
    kw["vectors_needed"] = int(os.environ["AB_VEC"])
params = m.ScanParams.from_config(W, H, **kw)
reps = (frames + distinct - 1) // distinct
counts = np.tile(np.diff(off.astype(np.int64)), reps)[:frames]
off_big = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
d_mv = torch.from_numpy(mv.view(np.uint8).copy()).to(dev).repeat(reps)[: int(off_big[-1]) * 40].contiguous()
d_off = torch.from_numpy(off_big).to(dev)
alg = 40 * int(off_big[-1]) + 9 * frames
new = m.MotionScanner(params, 0)
import ctypes as C  # noqa: E402
oldlib = C.CDLL(os.path.join(ROOT, "scripts", "libmtgpu_prev.so"))
for name, (res, args) in m._abi.ABI.items():      # an older build may lack the newest symbols
    if hasattr(oldlib, name):
        getattr(oldlib, name).restype = res
        getattr(oldlib, name).argtypes = args
orig = sc.load_library
sc.load_library = lambda: oldlib
try:
    old = m.MotionScanner(params, 0)
finally:
    sc.load_library = orig
variants = [("new", new, []), ("prev", old, []), ("new2", new, []), ("prev2", old, [])]
fl = torch.empty(frames, dtype=torch.uint8, device=dev)
for r in range(22):
    for name, s, times in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s.check_frames_device(d_mv, d_off, None, fl); e1.record(); torch.cuda.synchronize()
        if r >= 2:
            times.append(e0.elapsed_time(e1))
print("plan:", new.plan)
for name, s, times in variants:
    t = np.array(times)
    print(f"{wl} {frames:5d} {name:6s} median {np.median(t):.4f} ms  min {t.min():.4f}  {alg / np.median(t) / 1e6:7.0f} GB/s")
