#!/usr/bin/env python3
"""Round-5 probe: what making every n-th workgroup idle (as the frames without records of a stream do) is worth to a
kernel that only reads.  The calibration kernel of bench.py (mtgpu_debug_read_ceiling_shape, shape 2 = the scan's loads +
its arithmetic on a non-voting record, one workgroup per 1080p dense8x8 frame), 20 launches each, an event pair per launch,
on (a) a plain 20.7 GB buffer of random bytes and (b) the bench's own record buffer (bench.build_workload), with the settings
in two different orders.  Prints GB/s per setting."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
lib = m.load_library()
frame = 32640 * 40
st = torch.cuda.current_stream(dev).cuda_stream


def rate(ctx, buf, nbytes, idle, chunk=frame, shape=2):
    for _ in range(3):
        m._abi.check(lib.mtgpu_debug_read_ceiling_shape(ctx, buf.data_ptr(), nbytes, shape, chunk, idle, st))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record()
        m._abi.check(lib.mtgpu_debug_read_ceiling_shape(ctx, buf.data_ptr(), nbytes, shape, chunk, idle, st))
        b.record()
    torch.cuda.synchronize()
    return nbytes / (np.mean([a.elapsed_time(b) for a, b in evs]) * 1e-3) / 1e9


which = os.environ.get("PROBE_BUFFER", "bench")
if which == "bench":
    arena = torch.empty(bench.ARENA_BYTES, dtype=torch.uint8, device=dev)
    w = bench.build_workload("1080p_dense8x8", "code_defaults", 16384, 60, 1000, dev, arena)
    s, buf = w["scanner"], w["d_mv"]
    fl = w["d_flags"]
    for _ in range(5):
        s.check_frames_device(w["d_mv"], w["d_off"], None, fl)
else:
    s = m.MotionScanner(m.ScanParams.from_config(1920, 1080), 0)
    if os.environ.get("PROBE_SECOND_ALLOCATION") == "1":     # bench.py --no-others: the records are the process's SECOND 20 GB block
        first = torch.empty(15837 * frame, dtype=torch.uint8, device=dev)
        first.zero_()
    buf = torch.empty(15837 * frame, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    tile = torch.randint(0, 256, (60 * frame,), dtype=torch.uint8, device=dev, generator=g)
    buf[: (buf.numel() // tile.numel()) * tile.numel()].view(-1, tile.numel())[:] = tile
if os.environ.get("PROBE_SECOND_STREAM") == "1":          # what bench.py has and a bare probe has not: a second HIP stream that has run a kernel
    s2 = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s2):
        junk = torch.zeros(1024, device=dev) + 1
    s2.synchronize()
nbytes = (buf.numel() // 16) * 16
print(f"buffer: {which}, {nbytes} bytes at {buf.data_ptr():#x}")
for order in ((30, 0, 30, 0), (0, 120, 60, 30, 16, 8, 4, 0, 30)) if os.environ.get("PROBE_SHORT") == "1" else ((30, 0, 4, 8, 16, 30, 60, 120, 0), (0, 120, 60, 30, 16, 8, 4, 0, 30)):
    print(" | ".join(f"idle {i:3d}: {rate(s._ctx, buf, nbytes, i):7.1f}" for i in order), flush=True)
# (two more orders of the workgroups were probed with a kernel that is not kept — XCD-major: every XCD walks one contiguous
#  eighth of the buffer, 6660 against 6990 GB/s; octets of workgroups rotated by their index: profiles/r05_stagger_probe.txt)
print("1.25 MiB chunks: " + " | ".join(f"idle {i:3d}: {rate(s._ctx, buf, nbytes, i, 1280 * 1024):7.1f}" for i in (0, 30, 0, 30)), flush=True)
