#!/usr/bin/env python3
"""Round-5 probe, second step: regime_probe_r05.py showed that five scan + merge steps on two streams with event waits change
how the chip treats a launch in which every 30th workgroup is idle (6945 -> 7127 GB/s for the read-only kernel; the scan,
whose I-frames are such workgroups, 6.92 -> 7.1 TB/s).  Which part of that is it?  One action at a time."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = m.load_library()
s = m.MotionScanner(m.ScanParams.from_config(1920, 1080), 0)
frame = 32640 * 40
buf = torch.empty(15837 * frame, dtype=torch.uint8, device=dev)
buf.zero_()
nbytes = buf.numel()


def probe(tag, stream=None):
    st = (stream or torch.cuda.current_stream(dev)).cuda_stream
    out = []
    with torch.cuda.stream(stream or torch.cuda.current_stream(dev)):
        for idle in (30, 0):
            for _ in range(2):
                m._abi.check(lib.mtgpu_debug_read_ceiling_shape(s._ctx, buf.data_ptr(), nbytes, 2, frame, idle, st))
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for c0, c1 in evs:
                c0.record()
                m._abi.check(lib.mtgpu_debug_read_ceiling_shape(s._ctx, buf.data_ptr(), nbytes, 2, frame, idle, st))
                c1.record()
            torch.cuda.synchronize()
            out.append(f"idle {idle:2d}: {nbytes / (float(np.mean([c0.elapsed_time(c1) for c0, c1 in evs])) * 1e-3) / 1e9:.0f}")
    print(f"[{tag}] " + " | ".join(out), flush=True)


main = torch.cuda.current_stream(dev)
probe("0 fresh process, default stream")
s2 = torch.cuda.Stream(device=dev)
probe("1 a second stream exists")
with torch.cuda.stream(s2):
    x = torch.zeros(1 << 20, device=dev) + 1
s2.synchronize()
probe("2 ... and has run a kernel")
e = torch.cuda.Event()
e.record(main)
s2.wait_event(e)
s2.synchronize()
probe("3 the second stream has waited for an event of the default stream (no kernel behind it)")
e2 = torch.cuda.Event()
with torch.cuda.stream(s2):
    x += 1
    e2.record(s2)
main.wait_event(e2)
torch.cuda.synchronize()
probe("4 the default stream has waited for an event of the second stream")
for _ in range(5):
    m._abi.check(lib.mtgpu_debug_read_ceiling_shape(s._ctx, buf.data_ptr(), nbytes, 2, frame, 0, main.cuda_stream))
    e3 = torch.cuda.Event()
    e3.record(main)
    with torch.cuda.stream(s2):
        s2.wait_event(e3)
        x += 1
        e4 = torch.cuda.Event()
        e4.record(s2)
    main.wait_event(e4)
torch.cuda.synchronize()
probe("5 five rounds: big kernel on the default stream, small kernel on the second behind it, wait back")
probe("6 the same settings launched on the SECOND stream", s2)
s3 = torch.cuda.Stream(device=dev)
probe("7 ... on a third, fresh stream", s3)
