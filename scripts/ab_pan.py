#!/usr/bin/env python3
"""Vote-heavy ("pan": every record above the threshold) input, this build against another build of the library
(AB_OTHER_LIB, default scripts/libmtgpu_prev.so), interleaved in one process: HIP events around single calls.
Usage: AB_PAN=1 ab_pan.py workload params frames"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AB_PAN", "1")
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import scanner as sc  # noqa: E402

wl, pn, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
dev = torch.device("cuda", 0)
w = bench.build_workload(wl, pn, frames, 60, 1000, dev)
lib = C.CDLL(os.environ.get("AB_OTHER_LIB") or os.path.join(ROOT, "scripts", "libmtgpu_prev.so"))
for name, (res, args) in m._abi.ABI.items():
    if hasattr(lib, name):
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = args
orig = sc.load_library
sc.load_library = lambda: lib
try:
    other = m.MotionScanner(w["params"], 0)
finally:
    sc.load_library = orig
variants = [("this", w["scanner"], []), ("other", other, []), ("this2", w["scanner"], []), ("other2", other, [])]
fl = {n: torch.empty(frames, dtype=torch.uint8, device=dev) for n, _, _ in variants}
for r in range(int(os.environ.get("AB_ROUNDS", "10")) + 2):
    for name, s, times in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s.check_frames_device(w["d_mv"], w["d_off"], None, fl[name]); e1.record(); torch.cuda.synchronize()
        if r >= 2:
            times.append(e0.elapsed_time(e1))
assert all(torch.equal(fl["this"], fl[n]) for n in fl), "flags differ between the builds"
print("plan:", w["scanner"].plan, "motion frames", int(fl["this"].sum()))
for name, s, times in variants:
    t = np.array(times)
    print(f"{wl} {pn} {frames:5d} pan={os.environ['AB_PAN']} {name:7s} median {np.median(t):.4f} ms  {w['alg_bytes'] / np.median(t) / 1e6:7.0f} GB/s  frac {w['alg_bytes'] / np.median(t) / 1e6 / 8000:.4f}")
