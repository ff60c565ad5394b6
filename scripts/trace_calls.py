#!/usr/bin/env python3
"""What one scan CALL costs besides its scan kernel: run under `rocprofv3 --kernel-trace -f csv -d DIR -- python3
scripts/trace_calls.py run [workload frames]` (back-to-back calls on one stream, no host sync in between), then
`python3 scripts/trace_calls.py parse DIR` prints, per kernel name, the mean duration, and the mean gap from the end of
the previous kernel of the stream to this kernel's start."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(wl, frames, compact):
    import torch
    import bench
    import mvtrim_amd as m
    dev = torch.device("cuda", 0)
    w = bench.build_workload(wl, "code_defaults", frames, 60, 1000, dev)
    s = w["scanner"]
    if compact:
        import numpy as np
        rec = m.pack_records(w["mv"])
        d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev).repeat(w["reps"])[: w["n_records"] * 8].contiguous()
        call = lambda: s.check_frames_device_compact(d_rec, w["d_off"], None, w["d_flags"])
    else:
        call = lambda: s.check_frames_device(w["d_mv"], w["d_off"], None, w["d_flags"])
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    for _ in range(12):
        call()
    torch.cuda.synchronize()


def parse(d):
    rows = []
    for path in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        with open(path, newline="") as fh:
            rows += list(csv.DictReader(fh))
    rows = [r for r in rows if "mtgpu::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-36:] if len(rows) > 36 else rows          # the timed calls
    stats = {}
    prev_end = None
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void mtgpu::", "")[:48]
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        st = stats.setdefault(name, {"dur": [], "gap": []})
        st["dur"].append(b - a)
        if prev_end is not None:
            st["gap"].append(a - prev_end)
        prev_end = b
    for name, st in stats.items():
        dur = sum(st["dur"]) / len(st["dur"]) / 1e3
        gap = sum(st["gap"]) / max(len(st["gap"]), 1) / 1e3
        print(f"{name:50s} n {len(st['dur']):3d}  duration {dur:9.2f} us   gap before it {gap:7.2f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "1080p_dense8x8", int(sys.argv[3]) if len(sys.argv) > 3 else 16384,
            len(sys.argv) > 4 and sys.argv[4] == "compact")
    else:
        parse(sys.argv[2])
