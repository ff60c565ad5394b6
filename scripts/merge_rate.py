#!/usr/bin/env python3
"""Timing of merge_streams_kernel alone for various stream shapes (needs a GPU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
s = m.MotionScanner(m.ScanParams.from_config(1920, 1080))
rng = np.random.RandomState(0)
for (S, per, shuffle) in [(8, 512, False), (64, 512, False), (1, 4096, False), (1, 65536, False), (1, 1 << 20, False),
                          (8, 1 << 17, False), (1, 65536, True), (1, 1 << 20, True), (64, 16384, True)]:
    n = S * per
    flags = (rng.rand(n) < 0.3).astype(np.uint8)
    pts = np.concatenate([np.arange(per) / 30.0 for _ in range(S)])
    if shuffle:
        pts = np.concatenate([rng.permutation(np.arange(per) / 30.0) for _ in range(S)])
    off = np.arange(S + 1, dtype=np.int64) * per
    mp = np.concatenate([m.MergeParams(duration=per / 30.0).to_record() for _ in range(S)])
    d = [torch.from_numpy(x).to(dev) for x in (flags, pts, off, mp.view(np.uint8).copy())]
    out = (torch.zeros((S, 64, 2), dtype=torch.float64, device=dev), torch.zeros((S, 40), dtype=torch.uint8, device=dev),
           torch.empty(2 * n, dtype=torch.float64, device=dev))
    for _ in range(2):
        s.merge_streams_device(*d, True, 64, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 5
    e0.record()
    for _ in range(R):
        s.merge_streams_device(*d, True, 64, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"streams={S:3d} frames/stream={per:8d} shuffled={int(shuffle)}  {e0.elapsed_time(e1) / R * 1e3:10.1f} us")

# ---- one stream's pooled timestamps: single-workgroup kernel vs the multi-workgroup path
# (mtgpu_merge_timestamps_device picks by size; MTGPU_MERGE_LARGE_MIN moves the switch)
import json  # noqa: E402

rows = []
mp1 = m.MergeParams(duration=86400.0, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0)
scanners = {}
for name, lm in (("single_wg", str(1 << 30)), ("multi_wg", "1")):
    os.environ["MTGPU_MERGE_LARGE_MIN"] = lm
    scanners[name] = m.MotionScanner(m.ScanParams.from_config(1920, 1080))
del os.environ["MTGPU_MERGE_LARGE_MIN"]
for n in (256, 1024, 4096, 16384, 65536, 100_000, 1_000_000, 10_000_000):
    base = np.sort(rng.rand(n) * 86400.0)
    ts = np.round(base / 60.0) * 60.0 + rng.rand(n) * 20.0
    for order in ("sorted", "shuffled"):
        v = np.sort(ts) if order == "sorted" else rng.permutation(ts)
        d_ts = torch.from_numpy(v).to(dev)
        row = {"n": n, "order": order}
        for name, sc in scanners.items():
            if name == "single_wg" and n > 1_000_000:
                continue
            reps = 3 if (name == "single_wg" and n >= 65536) else 10
            for _ in range(2):
                seg, res = sc.merge_timestamps_device(d_ts, mp1, True, seg_cap=4096)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                seg, res = sc.merge_timestamps_device(d_ts, mp1, True, seg_cap=4096)
            e1.record()
            torch.cuda.synchronize()
            row[name + "_us"] = e0.elapsed_time(e1) / reps * 1e3
            row[name + "_segments"] = int(m.results_from_bytes(res.cpu().numpy()[None, :])[0]["n_segments"])
        rows.append(row)
        print(row, flush=True)
out = os.path.join(ROOT, "gpurun_out", "r02_merge_rate.json")
if os.path.isdir(os.path.dirname(out)):
    json.dump({"what": "mtgpu_merge_timestamps_device, one stream's pooled timestamps resident on the device, "
                       "HIP-event time per call (includes the stream-ordered workspace alloc/free)", "rows": rows},
              open(out, "w"), indent=1)
