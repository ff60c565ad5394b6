#!/usr/bin/env python3
"""What a scan CALL costs beyond its scan kernel: N back-to-back calls on one stream, wall clock between two device
synchronisations, for the library in the tree, scripts/libmtgpu_prev.so (the previous round: one workgroup per frame,
no planning kernels, no scratch for single-tile plans) and, when it is built, the experiments build.
Usage: call_cost.py [workload frames [compact]]
(The `exp+cache` rows of profiles/r06_call_cost.txt came from MTGPU_PLAN_CACHE=1 of an earlier build of the round: one
never-freed scratch block in place of the stream-ordered alloc / free pair — which the scratch ring replaced.)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import scanner as sc  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "1080p_dense8x8"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
compact = len(sys.argv) > 3 and sys.argv[3] == "compact"
dev = torch.device("cuda", 0)
w = bench.build_workload(wl, "code_defaults", frames, 60, 1000, dev)
params = w["params"]


def other(path, env=None):
    lib = C.CDLL(path)
    for name, (res, args) in m._abi.ABI.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
    orig = sc.load_library
    sc.load_library = lambda: lib
    for k_, v_ in (env or {}).items():
        os.environ[k_] = v_
    try:
        return m.MotionScanner(params, 0)
    finally:
        sc.load_library = orig
        for k_ in (env or {}):
            os.environ.pop(k_, None)


builds = [("new", w["scanner"])]
prev = os.path.join(ROOT, "scripts", "libmtgpu_prev.so")
exp = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "libmtgpu_experiments.so")
if os.path.exists(prev):
    builds.append(("prev", other(prev)))
if os.path.exists(exp):
    builds.append(("exp", other(exp)))
if compact:
    rec = m.pack_records(w["mv"])
    d_in = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev).repeat(w["reps"])[: w["n_records"] * 8].contiguous()
else:
    d_in = w["d_mv"]
N = int(os.environ.get("CALLS", "100"))
res = {name: [] for name, _ in builds}
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
base = np.median(res["prev"]) if "prev" in res else None
for name, _ in builds:
    t = np.median(res[name])
    print(f"{wl} {frames} {'compact' if compact else 'aos40'} {name:10s} {t:9.2f} us per call"
          + (f"  ({t - base:+7.2f} us vs prev)" if base is not None else ""), flush=True)
