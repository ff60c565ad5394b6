#!/usr/bin/env python3
"""Round-5 A/B of how banded plans write and replay their spill queue (scan_kernels.hip: DEFER / PIPE / DROP / DEEP),
all variants interleaved in ONE process on the same resident batch.  HISTORICAL: the variants exist only in the
experiments build of commit 00598d2 (git checkout 00598d2 -- motion-estimated-video-trimmer_amd/csrc); they measured equal
and were removed again when the span queue went in (c9d1a87).  Logs: profiles/r05_ab_spill_*.log.  Needs that build:

    make -C motion-estimated-video-trimmer_amd/csrc experiments
    MTGPU_LIBRARY=$PWD/motion-estimated-video-trimmer_amd/libmtgpu_experiments.so python scripts/ab_spill_r05.py

MTGPU_VARIANT = 128 | bits: 8 PIPE, 16 DEFER, 32 DROP, 64 DEEP, 256 BATCH (4-bit thermometer form on 40-byte records only).
Every variant's flags must equal variant 0's.  One line per (case, variant)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import mvtrim_amd as m  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [  # workload, params, frames, pan
    ("4k_fine_dense4", "shipped_env", 256, True),      # 4 records per 4x4 block, every one votes
    ("4k_fine", "shipped_env", 1024, True),            # one record per block, every one votes (no runs to find)
    ("4k_fine_dense4", "shipped_env", 1024, False),    # the bench leg: typical input on the same plan
    ("4k_fine_dense4", "shipped_env", 1024, True),     # the profiled pan row (profiles/pmc_traffic.json)
]
VARIANTS = [int(v) for v in os.environ.get("AB_VARIANTS", "0,256,264,288,296,320,360,56").split(",")]
ROUNDS = int(os.environ.get("AB_ROUNDS", "7"))
only = os.environ.get("ONLY")
for ci, (wl, pn, frames, pan) in enumerate(CASES):
    if only and str(ci) not in only.split(","):
        continue
    os.environ["AB_PAN"] = "1" if pan else "0"
    os.environ.pop("MTGPU_VARIANT", None)
    w = bench.build_workload(wl, pn, frames, 30, 1000, dev)
    w["scanner"].close()
    scanners = []
    for v in VARIANTS:
        os.environ["MTGPU_VARIANT"] = str(128 | v)
        scanners.append((v, m.MotionScanner(w["params"], 0), torch.empty(frames, dtype=torch.uint8, device=dev), []))
    os.environ.pop("MTGPU_VARIANT", None)
    ref = None
    for r in range(ROUNDS + 1):
        for v, s, fl, times in scanners:
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
            for e0, e1 in evs:                      # four launches back to back, as the bench queues them
                e0.record()
                s.check_frames_device(w["d_mv"], w["d_off"], None, fl)
                e1.record()
            torch.cuda.synchronize()
            if r >= 1:
                times.extend(e0.elapsed_time(e1) for e0, e1 in evs)
            if ref is None:
                ref = fl.clone()
            assert (v & (512 | 1024 | 2048)) or torch.equal(fl, ref), f"variant {v} disagrees with variant {VARIANTS[0]}"
    for v, s, fl, times in scanners:
        t = np.array(times)
        bits = "+".join(n for b, n in ((8, "PIPE"), (16, "DEFER"), (32, "DROP"), (64, "DEEP"), (256, "BATCH"), (512, "nostore"), (1024, "novote"), (2048, "noreplay"), (4096, "nt"), (8192, "sc1")) if v & b) or "round4"
        print(f"case {ci} {wl}:{pn}:{frames}{':pan' if pan else ''} var {v:3d} {bits:22s} bands {s.plan['bands']} fb {s.plan['counter_bits']} "
              f"median {np.median(t):.4f} ms min {t.min():.4f}  {w['alg_bytes'] / np.median(t) / 1e9:.3f} TB/s "
              f"frac {w['alg_bytes'] / np.median(t) / 1e9 / 8.0:.3f} motion {int(ref.sum())}/{frames}", flush=True)
        s.close()
    del w, scanners
    torch.cuda.empty_cache()
