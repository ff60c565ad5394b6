#!/usr/bin/env python3
"""Scan throughput on ragged batches (frames with very different record counts) — checks
that one-workgroup-per-frame dispatch stays balanced.  Needs a GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
spec = synth.spec_1080p(seed=5)
spec.events = synth.scripted_events(spec, 31)
mv, off, pts, sd = synth.gen_stream(spec, 31)
tile = torch.from_numpy(mv.view(np.uint8).copy()).to(dev)
nrec_tile = len(mv)
s = m.MotionScanner(m.ScanParams.from_config(1920, 1080))
rng = np.random.RandomState(1)
for name, gen in [("uniform 32640", lambda n: np.full(n, 32640)),
                  ("uniform[1000,32640]", lambda n: rng.randint(1000, 32641, size=n)),
                  ("bimodal 2000 / 32640", lambda n: np.where(rng.rand(n) < 0.5, 2000, 32640)),
                  ("heavy tail (5% x 300k)", lambda n: np.where(rng.rand(n) < 0.05, 300000, 8160)),
                  ("tiny 500", lambda n: np.full(n, 500))]:
    frames = 8192
    counts = gen(frames).astype(np.int64)
    total = int(counts.sum())
    reps = (total + nrec_tile - 1) // nrec_tile
    d_mv = tile.repeat(reps)[: total * 40].contiguous()
    d_off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)])).to(dev)
    fl = torch.empty(frames, dtype=torch.uint8, device=dev)
    for _ in range(2):
        s.check_frames_device(d_mv, d_off, None, fl)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s.check_frames_device(d_mv, d_off, None, fl); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts))
    print(f"{name:26s} frames={frames} bytes={total * 40 / 1e9:6.2f} GB  {t:7.3f} ms  {total * 40 / t / 1e6:7.0f} GB/s  {frames / t * 1e3 / 1e6:6.2f} M frames/s")
    del d_mv
