#!/usr/bin/env python3
"""Scan rate against the key-frame period of the stream (frames WITHOUT side data every P frames), and on a ragged
stream: the same records, cut into frames differently, scanned by up to three builds interleaved in ONE process —
    new    the library in the tree (work list: frames without side data never get a workgroup)
    prev   scripts/libmtgpu_prev.so, built from the previous round's commit (one workgroup per frame), when it exists
    exp:K  motion-estimated-video-trimmer_amd/libmtgpu_experiments.so with MTGPU_RESIDENT=K (ticketed resident
           workgroups, K per CU, over the same work list), for every K in GOP_RESIDENT (e.g. "2,4")
Usage: gop_sweep.py [frames_with_records] [periods, comma separated; 0 = no key frames; r = ragged]
The flags of every build are compared frame by frame for every case."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402
from bench import make_spec  # noqa: E402
from mvtrim_amd import synth, scanner as sc  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 15872
periods = (sys.argv[2] if len(sys.argv) > 2 else "0,2,8,12,15,16,25,30,32,50,250,r").split(",")
wl = os.environ.get("GOP_WORKLOAD", "1080p_dense8x8")
rounds = int(os.environ.get("GOP_ROUNDS", "12"))
dev = torch.device("cuda", 0)
spec, (W, H, gridkw) = make_spec(wl, seed=1000)
spec.gop = 0                                   # the generator makes frames WITH records only; key frames are cut in below
distinct = 60
spec.events = synth.scripted_events(spec, distinct)
mv, off, pts, sd = synth.gen_stream(spec, distinct)
kw = dict(m.config.CODE_DEFAULTS)
kw.update(gridkw)
if spec.sub == 1:
    kw["vectors_needed"] = 1
params = m.ScanParams.from_config(W, H, **kw)
reps = (M + distinct - 1) // distinct
counts = np.tile(np.diff(off.astype(np.int64)), reps)[:M]
n_records = int(counts.sum())
compact = os.environ.get("GOP_COMPACT", "0") == "1"        # the 8-byte compact records of the host dispatcher instead
REC = 8 if compact else 40
tile = m.pack_records(mv).view(np.uint8).reshape(-1) if compact else mv.view(np.uint8)
d_mv = torch.from_numpy(tile.copy()).to(dev).repeat(reps)[: n_records * REC].contiguous()


def other_library(path, env=None):
    lib = C.CDLL(path)
    for name, (res, args) in m._abi.ABI.items():      # another build may lack (or add) symbols
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
    orig = sc.load_library
    sc.load_library = lambda: lib
    for k_, v_ in (env or {}).items():
        os.environ[k_] = v_
    try:
        return m.MotionScanner(params, 0)
    finally:
        sc.load_library = orig
        for k_ in (env or {}):
            os.environ.pop(k_, None)


builds = [("new", m.MotionScanner(params, 0))]
for spec_ in [x for x in os.environ.get("GOP_GROUPS", "").split(",") if x]:       # e.g. "1,2,8": MTGPU_GROUP (a trailing "p" once forced the work list)
    os.environ["MTGPU_GROUP"] = spec_.rstrip("p")
    builds.append((f"g{spec_}", m.MotionScanner(params, 0)))
    os.environ.pop("MTGPU_GROUP")
prev = os.environ.get("GOP_PREV_LIB") or os.path.join(ROOT, "scripts", "libmtgpu_prev.so")
if os.path.exists(prev) and os.environ.get("GOP_PREV", "1") != "0":
    builds.append(("prev", other_library(prev)))
exp = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "libmtgpu_experiments.so")
for k_ in [x for x in os.environ.get("GOP_RESIDENT", "").split(",") if x]:
    if os.path.exists(exp):
        builds.append((f"exp:{k_}", other_library(exp, {"MTGPU_RESIDENT": k_})))
print("plan:", builds[0][1].plan, "| records", n_records, "| frames with records", M, flush=True)


def frame_counts(period):
    if period == "r":                          # ragged: the same record array cut at random points (0.1 .. 1.9 x a frame)
        rng = np.random.RandomState(5)
        avg = n_records // M
        c = rng.randint(avg // 10, avg * 19 // 10, size=M).astype(np.int64)
        c[rng.rand(M) < 0.03] = 0              # and 3 % of the frames without records, at random places
        c = c[: int(np.searchsorted(np.cumsum(c), n_records, side="right"))]      # frames that fit the record array
        return c
    p = int(period)
    if p <= 0:
        return counts
    if p == 1:
        raise SystemExit("period 1: every frame a key frame")
    out = []
    for i in range(0, M, p - 1):               # a key frame, then p - 1 frames with records
        out.append(np.zeros(1, dtype=np.int64))
        out.append(counts[i:i + p - 1])
    return np.concatenate(out)


base = {}
for period in periods:
    c = frame_counts(period)
    F = len(c)
    off_big = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
    nrec = int(off_big[-1])
    d_off = torch.from_numpy(off_big).to(dev)
    alg = REC * nrec + 9 * F
    times = {name: [] for name, _ in builds}
    flags = {name: torch.empty(F, dtype=torch.uint8, device=dev) for name, _ in builds}
    for r in range(rounds + 2):
        for name, s in builds:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            (s.check_frames_device_compact if compact else s.check_frames_device)(d_mv[: nrec * REC], d_off, None, flags[name])
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[name].append(e0.elapsed_time(e1))
    ref = flags[builds[0][0]].cpu().numpy()
    for name, _ in builds[1:]:
        assert np.array_equal(flags[name].cpu().numpy(), ref), (period, name, "flags differ between builds")
    row = f"period {period:>4s} frames {F:6d} (empty {int((c == 0).sum()):5d})"
    for name, _ in builds:
        t = np.array(times[name])
        rate = alg / np.median(t) / 1e6
        base.setdefault(name, rate if period == "0" else None)
        rel = f" ({rate / base[name] - 1:+.1%})" if base.get(name) else ""
        row += f" | {name} {np.median(t):.4f} ms {rate:6.0f} GB/s{rel}"
    print(row, flush=True)
