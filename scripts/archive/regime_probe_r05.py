#!/usr/bin/env python3
"""Round-5 probe: inside bench.py the read-only calibration kernel is FASTER with every 30th workgroup idle (7100-7160 vs
7035-7080 GB/s), in a bare process SLOWER (6900-7000 vs 7000-7090) — scripts/stagger_probe_r05.py.  This script walks
through what bench.py does before its timed loop, one action at a time, and measures both settings after each, to see
which action changes the sign."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = m.load_library()


def probe(tag, scanner, d_mv):
    st = torch.cuda.current_stream(dev).cuda_stream
    nbytes = (d_mv.numel() // 16) * 16
    out = []
    for idle in (30, 0, 30, 0):
        for _ in range(3):
            m._abi.check(lib.mtgpu_debug_read_ceiling_shape(scanner._ctx, d_mv.data_ptr(), nbytes, 2, 1305600, idle, st))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for c0, c1 in evs:
            c0.record()
            m._abi.check(lib.mtgpu_debug_read_ceiling_shape(scanner._ctx, d_mv.data_ptr(), nbytes, 2, 1305600, idle, st))
            c1.record()
        torch.cuda.synchronize()
        out.append(f"idle {idle:2d}: {nbytes / (float(np.mean([c0.elapsed_time(c1) for c0, c1 in evs])) * 1e-3) / 1e9:.0f}")
    print(f"[{tag}] " + " | ".join(out), flush=True)


frames = 16384
w = bench.build_workload("1080p_dense8x8", "code_defaults", frames, 60, 1000, dev, None)
spec, scanner, d_mv, d_off, d_flags = w["spec"], w["scanner"], w["d_mv"], w["d_off"], w["d_flags"]
probe("1 after build_workload", scanner, d_mv)
S, per = 8, frames // 8
stream_off = np.array([i * per for i in range(S)] + [frames], dtype=np.int64)
pts_big = np.concatenate([np.array([spec.pts_seconds(i) for i in range(per)]) for _ in range(S)])
mp = np.concatenate([m.MergeParams(duration=per / spec.fps).to_record() for _ in range(S)])
d_pts, d_soff = torch.from_numpy(pts_big).to(dev), torch.from_numpy(stream_off).to(dev)
d_mp = torch.from_numpy(mp.view(np.uint8).copy()).to(dev)
outs = (torch.zeros((S, 64, 2), dtype=torch.float64, device=dev), torch.zeros((S, m.MERGE_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev),
        torch.empty(2 * frames, dtype=torch.float64, device=dev))
probe("2 after the merge inputs / outputs were allocated", scanner, d_mv)
scanner.check_frames_device(d_mv, d_off, None, d_flags)
torch.cuda.synchronize()
def scan_rate(tag):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for c0, c1 in evs:
        c0.record()
        scanner.check_frames_device(d_mv, d_off, None, d_flags)
        c1.record()
    torch.cuda.synchronize()
    print(f"scan itself [{tag}]: {w['alg_bytes'] / (float(np.mean([c0.elapsed_time(c1) for c0, c1 in evs])) * 1e-3) / 1e9:.0f} GB/s", flush=True)


scan_rate("one stream so far")
probe("3 after 21 scan launches", scanner, d_mv)
scanner.merge_streams_device(d_flags, d_pts, d_soff, d_mp, True, 64, out=outs)
torch.cuda.synchronize()
probe("4 after one merge launch on the same stream", scanner, d_mv)
merge_stream = torch.cuda.Stream(device=dev)
scan_stream = torch.cuda.current_stream(dev)
done = torch.cuda.Event()
for _ in range(5):
    scanner.check_frames_device(d_mv, d_off, None, d_flags)
    done.record(scan_stream)
    with torch.cuda.stream(merge_stream):
        merge_stream.wait_event(done)
        scanner.merge_streams_device(d_flags, d_pts, d_soff, d_mp, True, 64, out=outs)
        back = torch.cuda.Event()
        back.record(merge_stream)
    scan_stream.wait_event(back)
torch.cuda.synchronize()
probe("5 after five scan + merge steps on two streams with event waits", scanner, d_mv)
scan_rate("after the two-stream steps")
probe("6 after 20 more scans", scanner, d_mv)
