#!/usr/bin/env python3
"""Round-5 probe: what staggering the workgroups of a launch is worth to a kernel that only reads.  The calibration kernel
of bench.py (mtgpu_debug_read_ceiling_shape, shape 2 = the scan's loads + its arithmetic on a non-voting record, one
workgroup per 1080p dense8x8 frame) with every n-th workgroup idle, n = 0 (none) ... 120, on a 20.7 GB buffer, 20 launches each,
an event pair per launch.  Prints GB/s per setting, three passes."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
lib = m.load_library()
s = m.MotionScanner(m.ScanParams.from_config(1920, 1080), 0)
frame = 32640 * 40
buf = torch.empty(15837 * frame, dtype=torch.uint8, device=dev)
if os.environ.get("PROBE_FILL", "random") == "zero":
    buf.zero_()
else:                                   # record-like content: random bytes
    g = torch.Generator(device=dev); g.manual_seed(1)
    tile = torch.randint(0, 256, (60 * frame,), dtype=torch.uint8, device=dev, generator=g)
    buf.view(-1)[: (buf.numel() // tile.numel()) * tile.numel()].view(-1, tile.numel())[:] = tile
st = torch.cuda.current_stream(dev).cuda_stream
for rnd in range(3):
    out = []
    for idle in (30, 0, 4, 8, 16, 30, 60, 120, 0):
        for _ in range(3):
            m._abi.check(lib.mtgpu_debug_read_ceiling_shape(s._ctx, buf.data_ptr(), buf.numel(), 2, frame, idle, st))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record()
            m._abi.check(lib.mtgpu_debug_read_ceiling_shape(s._ctx, buf.data_ptr(), buf.numel(), 2, frame, idle, st))
            b.record()
        torch.cuda.synchronize()
        out.append(f"idle every {idle:3d}: {buf.numel() / (np.mean([a.elapsed_time(b) for a, b in evs]) * 1e-3) / 1e9:7.1f} GB/s")
    print(" | ".join(out), flush=True)
