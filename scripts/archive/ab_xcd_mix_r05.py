#!/usr/bin/env python3
"""Round-5 A/B: the hardware deals the workgroups of a launch to its 8 XCDs in turn, so frames without records (I-frames:
workgroups that leave at once) with a period that shares a factor with 8 fall on few XCDs and leave those idle.  MTGPU_XCD_MIX=1
(experiments build) rotates the 8 items of every octet of workgroups by a hash of the octet's index.  1080p dense8x8, 16 384
frames, I-frame period 8 ... 250 and none, both mappings interleaved in one process, in the state bench.py runs in (another
stream has run a kernel).  Needs MTGPU_LIBRARY=.../libmtgpu_experiments.so."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
orig = bench.make_spec
frames = int(os.environ.get("AB_FRAMES", "16384"))
wl = os.environ.get("AB_WORKLOAD", "1080p_dense8x8")
for gop in (30, 8, 16, 32, 60, 15, 12, 250, 0):
    def make_spec(workload, seed, _g=gop):
        spec, rest = orig(workload, seed)
        spec.gop = _g
        return spec, rest
    bench.make_spec = make_spec
    os.environ.pop("MTGPU_XCD_MIX", None)
    w = bench.build_workload(wl, "code_defaults", frames, 240, 1000, dev)
    os.environ["MTGPU_XCD_MIX"] = "1"
    mixed = m.MotionScanner(w["params"], 0)
    os.environ.pop("MTGPU_XCD_MIX", None)
    s2 = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s2):
        x = torch.zeros(1 << 20, device=dev) + 1
    s2.synchronize()
    res = {"identity": [], "mixed": []}
    fl = {"identity": w["d_flags"], "mixed": torch.empty_like(w["d_flags"])}
    for r in range(4):
        for name, sc in (("identity", w["scanner"]), ("mixed", mixed)):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
            for a, b in evs:
                a.record()
                sc.check_frames_device(w["d_mv"], w["d_off"], None, fl[name])
                b.record()
            torch.cuda.synchronize()
            if r:
                res[name] += [a.elapsed_time(b) for a, b in evs]
    assert torch.equal(fl["identity"], fl["mixed"])
    print(f"{wl} I-frame period {gop:3d}: " + " | ".join(f"{n} {w['alg_bytes'] / (np.mean(t) * 1e-3) / 1e9:7.1f} GB/s" for n, t in res.items()), flush=True)
    w["scanner"].close()
    mixed.close()
    del w
    torch.cuda.empty_cache()
