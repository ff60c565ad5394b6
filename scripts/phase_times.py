#!/usr/bin/env python3
"""Where a scan workgroup's time goes (developer tool).  Needs the instrumented build:
    make -C motion-estimated-video-trimmer_amd/csrc clean all EXTRA=-DMTGPU_PHASE_TIMES
Usage: python scripts/phase_times.py [workload frames vec slices] ...   e.g.  4k_fine 1024 4 1
(a workload name with the suffix ":c" scans the batch as 8-byte compact records)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
lib = m.load_library()
if not hasattr(lib, "mtgpu_debug_set_phase_times"):
    raise SystemExit("libmtgpu.so was not built with EXTRA=-DMTGPU_PHASE_TIMES")
lib.mtgpu_debug_set_phase_times.argtypes = [C.c_void_p]
args = sys.argv[1:] or ["4k_fine", "1024", "1", "1"]
cases = [args[i:i + 4] for i in range(0, len(args), 4)]
for wl, frames, vec, S in cases:
    compact = wl.endswith(":c")
    wl = wl[:-2] if compact else wl
    frames, vec, S = int(frames), int(vec), int(S)
    os.environ["AB_VEC"] = str(vec)
    w = bench.build_workload(wl, "code_defaults", frames, 30, 1000, dev)
    if vec != w["params"].vectors_needed:
        w["scanner"].close()
        p = w["params"]
        p.vectors_needed = vec
        w["scanner"] = m.MotionScanner(p, 0)
    s = w["scanner"]
    s.set_slices(S)
    plan = s.plan
    items = frames * max(S, 1)
    if compact:
        rec = m.pack_records(w["mv"])
        d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev).repeat(w["reps"])[: w["n_records"] * 8].contiguous()

        def launch():
            s.check_frames_device_compact(d_rec, w["d_off"], None, w["d_flags"])
    else:
        def launch():
            s.check_frames_device(w["d_mv"], w["d_off"], None, w["d_flags"])
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    buf = torch.zeros(items * 8, dtype=torch.int64, device=dev)
    assert lib.mtgpu_debug_set_phase_times(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    assert lib.mtgpu_debug_set_phase_times(None) == 0
    t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64) * 0.01          # microseconds
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()

    def st(x):
        return "mean %7.1f  p50 %7.1f  p95 %7.1f  max %7.1f" % (x.mean(), np.median(x), np.percentile(x, 95), x.max())
    print(f"{wl}{' (compact records)' if compact else ''} frames={frames} vec={vec} slices={S} plan: fb={plan['counter_bits']} block={plan['block_threads']} "
          f"bands={plan['bands']} lds={plan['lds_bytes']}  kernel {e0.elapsed_time(e1):.3f} ms, {len(t)} workgroups with work")
    print("  start offset  :", st(t[:, 0] - t0), " concurrent ~%.0f" % ((t[:, 1] - t[:, 0]).sum() / (t[:, 1].max() - t0)))
    print("  lifetime      :", st(t[:, 1] - t[:, 0]))
    for name, c in (("zero", 2), ("stream", 3), ("queue replay", 4), ("slice hand-off", 5), ("cluster test", 6)):
        print(f"  {name:14s}:", st(t[:, c]))
    w["scanner"].close()
    del w, buf
    torch.cuda.empty_cache()
