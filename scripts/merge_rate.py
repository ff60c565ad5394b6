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
