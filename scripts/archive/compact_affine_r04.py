#!/usr/bin/env python3
"""is the compact (8-byte record) kernel's shortfall at small batches a launch ramp / tail?
Kernel time by HIP events for 1080p dense8x8 at 1024 ... 32768 frames per launch, compact and 40-byte records,
then a least-squares line t = a + b * frames: `a` = fixed cost per launch, 1 / b = marginal rate.
Prints JSON (keep under profiles/).  Needs a GPU."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
rows = []
for frames in (1024, 2048, 4096, 8192, 16384, 32768):
    w = bench.build_workload("1080p_dense8x8", "code_defaults", frames, 30, 1000, dev)
    k40 = bench.time_scan_only(w, 30)
    k8, _ = bench.time_compact(w, 30)
    cb = 8 * w["n_records"] + 9 * frames
    rows.append({"frames": frames, "aos40_ms": k40, "aos40_TBps": w["alg_bytes"] / k40 / 1e9,
                 "compact_ms": k8, "compact_TBps_of_compact_bytes": cb / k8 / 1e9, "compact_bytes": cb,
                 "aos40_bytes": w["alg_bytes"]})
    print(rows[-1], file=sys.stderr, flush=True)
    w["scanner"].close()
    del w
    torch.cuda.empty_cache()
out = {"workload": "1080p dense8x8, code defaults, 30 distinct frames tiled", "rows": rows}
x = np.array([r["frames"] for r in rows], dtype=np.float64)
for key, bkey in (("compact_ms", "compact_bytes"), ("aos40_ms", "aos40_bytes")):
    y = np.array([r[key] for r in rows])
    b, a = np.polyfit(x, y, 1)
    per_frame_bytes = rows[-1][bkey] / rows[-1]["frames"]
    resid = y - (a + b * x)
    out[key + "_fit"] = {"fixed_ms_per_launch": a, "marginal_ms_per_frame": b,
                         "marginal_TBps": per_frame_bytes / b / 1e9, "max_abs_residual_ms": float(np.abs(resid).max()),
                         "affine": bool(np.abs(resid).max() < 0.03 * y.max())}
print(json.dumps(out, indent=1))
