#!/usr/bin/env python3
"""Interleaved A/B of scan-kernel launch plans in ONE process (guide rule 24): for each
workload build one scanner per variant (MTGPU_FORCE_FB / MTGPU_FORCE_BLOCK), run the
variants round-robin, report median / min kernel time and algorithmic GB/s.
Usage: python scripts/ab_scan.py [workload ...]   (needs a GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mvtrim_amd as m  # noqa: E402
from bench import make_spec  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

VARIANTS_FINE = [("default", dict()), ("chunk64", dict(MTGPU_FORCE_CHUNK="64")), ("chunk128", dict(MTGPU_FORCE_CHUNK="128")),
                 ("chunk64/b1024", dict(MTGPU_FORCE_CHUNK="64", MTGPU_FORCE_BLOCK="1024")),
                 ("chunk32/b512", dict(MTGPU_FORCE_CHUNK="32", MTGPU_FORCE_BLOCK="512")),
                 ("chunk160/b512", dict(MTGPU_FORCE_CHUNK="160", MTGPU_FORCE_BLOCK="512")),
                 ("fb2", dict(MTGPU_FORCE_FB="2"))]
VARIANTS_KERNEL = [("v%d" % v, dict(MTGPU_VARIANT=str(v))) for v in (0, 1, 2, 4, 8, 9, 10, 12, 3, 11, 6, 14)]
VARIANTS_SLICES = [("auto", dict()), ("s1", dict(MTGPU_FORCE_SLICES="1")), ("s2", dict(MTGPU_FORCE_SLICES="2")),
                   ("s4", dict(MTGPU_FORCE_SLICES="4")), ("s8", dict(MTGPU_FORCE_SLICES="8"))]
# row bands walked by one workgroup (spill queue): tile size / workgroup size (use with AB_VEC=4 on 4k_fine)
VARIANTS_BANDS = [("auto", dict()), ("tile160", dict(MTGPU_BAND_LDS_KB="160")), ("tile160/b1024", dict(MTGPU_BAND_LDS_KB="160", MTGPU_FORCE_BLOCK="1024")),
                  ("tile120/b1024", dict(MTGPU_BAND_LDS_KB="120", MTGPU_FORCE_BLOCK="1024")), ("tile80/b512", dict(MTGPU_FORCE_BLOCK="512")),
                  ("tile80/b1024", dict(MTGPU_FORCE_BLOCK="1024")),
                  ("tile53", dict(MTGPU_BAND_LDS_KB="53")), ("tile53/b512", dict(MTGPU_BAND_LDS_KB="53", MTGPU_FORCE_BLOCK="512")),
                  ("tile40/b512", dict(MTGPU_BAND_LDS_KB="40", MTGPU_FORCE_BLOCK="512")),
                  ("tile120", dict(MTGPU_BAND_LDS_KB="120"))]
# one 160 KB tile vs two spill bands of <= 80 KB (use with AB_VEC=2 on 4k_fine)
VARIANTS_TILE = [("single", dict()), ("bands80", dict(MTGPU_MAX_TILE_KB="80")),
                 ("bands80/b512", dict(MTGPU_MAX_TILE_KB="80", MTGPU_FORCE_BLOCK="512")),
                 ("bands53", dict(MTGPU_MAX_TILE_KB="53", MTGPU_BAND_LDS_KB="53"))]
VARIANTS_GROUP = [("auto", dict()), ("g1", dict(MTGPU_GROUP="1")), ("g2", dict(MTGPU_GROUP="2")), ("g4", dict(MTGPU_GROUP="4")),
                  ("g8", dict(MTGPU_GROUP="8"))]
# compact records: frames per workgroup x next-frame prefetch (use with AB_COMPACT=1)
VARIANTS_PREFETCH = [("g1", dict(MTGPU_GROUP="1")), ("g2/pf", dict(MTGPU_GROUP="2")), ("g2/nopf", dict(MTGPU_GROUP="2", MTGPU_PREFETCH="0")),
                     ("g4/pf", dict(MTGPU_GROUP="4")), ("g4/nopf", dict(MTGPU_GROUP="4", MTGPU_PREFETCH="0")),
                     ("g8/pf", dict(MTGPU_GROUP="8")), ("auto", dict())]
# 40-byte records: streams start on a 128-byte line (head records peeled) or wherever the frame starts
VARIANTS_ALIGN = [("aligned", dict()), ("unaligned", dict(MTGPU_ALIGN="0")), ("aligned2", dict()), ("unaligned2", dict(MTGPU_ALIGN="0"))]
# workgroups per CU, capped through the LDS size of the launch (32-bit counters on a 1080p / 4K grid)
VARIANTS_OCC = [("auto", dict()), ("lds40", dict(MTGPU_MIN_LDS_KB="40")), ("lds53", dict(MTGPU_MIN_LDS_KB="53")),
                ("lds80", dict(MTGPU_MIN_LDS_KB="80")), ("lds160", dict(MTGPU_MIN_LDS_KB="160"))]
VARIANTS = [("fb32", dict(MTGPU_FORCE_FB="32")), ("fb2", dict(MTGPU_FORCE_FB="2")),
            ("fb32/b512", dict(MTGPU_FORCE_FB="32", MTGPU_FORCE_BLOCK="512")),
            ("fb2/b512", dict(MTGPU_FORCE_FB="2", MTGPU_FORCE_BLOCK="512")),
            ("fb2/b1024", dict(MTGPU_FORCE_FB="2", MTGPU_FORCE_BLOCK="1024")),
            ("fb32/b1024", dict(MTGPU_FORCE_FB="32", MTGPU_FORCE_BLOCK="1024"))]
FRAMES = {"1080p_dense8x8": 4096, "1080p_dense16": 16384, "4k_dense8x8": 1024, "4k_fine": 256, "4k_fine_dense4": 1024, "4k_dense16": 16384,
          "480p_dense16": 262144, "480p_dense8x8": 65536, "720p_dense16": 131072, "720p_dense8x8": 32768}


def main():
    dev = torch.device("cuda", 0)
    rounds = int(os.environ.get("AB_ROUNDS", "15"))
    pan = os.environ.get("AB_PAN", "0") == "1"      # every MV above threshold (camera pan): vote-path stress
    for wl in (sys.argv[1:] or list(FRAMES)):
        spec, (W, H, gridkw) = make_spec(wl, seed=1)
        distinct = 30
        spec.events = synth.scripted_events(spec, distinct)
        if pan:
            spec.events = [synth.Event(0, distinct, 0, 0, spec.cells_x, spec.cells_y, 9, 3)]
        if os.environ.get("AB_QUIET") == "1":       # no record passes the threshold: the vote path never runs
            spec.events, spec.salt_p, spec.oob_p = [], 0.0, 0.0
        mv, off, pts, sd = synth.gen_stream(spec, distinct)
        kw = dict(m.config.CODE_DEFAULTS)
        kw.update(gridkw)
        if spec.sub == 1:
            kw["vectors_needed"] = 1
        if os.environ.get("AB_VEC"):
            kw["vectors_needed"] = int(os.environ["AB_VEC"])
        params = m.ScanParams.from_config(W, H, **kw)
        frames = int(os.environ.get("AB_FRAMES", FRAMES[wl]))
        reps = (frames + distinct - 1) // distinct
        counts = np.tile(np.diff(off.astype(np.int64)), reps)[:frames]
        off_big = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        d_mv = torch.from_numpy(mv.view(np.uint8).copy()).to(dev).repeat(reps)[: int(off_big[-1]) * 40].contiguous()
        d_off = torch.from_numpy(off_big).to(dev)
        alg = 40 * int(off_big[-1]) + 9 * frames
        compact = os.environ.get("AB_COMPACT") == "1"
        if compact:                                 # the same batch as 8-byte compact records
            rec = m.pack_records(mv)
            d_mv = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev).repeat(reps)[: int(off_big[-1]) * 8].contiguous()
            alg = 8 * int(off_big[-1]) + 9 * frames
        scanners = []
        vset = {"fine": VARIANTS_FINE, "kernel": VARIANTS_KERNEL, "slices": VARIANTS_SLICES,
                "bands": VARIANTS_BANDS, "tile": VARIANTS_TILE, "group": VARIANTS_GROUP, "occ": VARIANTS_OCC,
                "prefetch": VARIANTS_PREFETCH, "align": VARIANTS_ALIGN}.get(os.environ.get("AB_SET"), VARIANTS)
        for name, env in vset:
            for k in ("MTGPU_FORCE_FB", "MTGPU_FORCE_BLOCK", "MTGPU_FORCE_CHUNK", "MTGPU_VARIANT", "MTGPU_FORCE_SLICES",
                      "MTGPU_BAND_LDS_KB", "MTGPU_MAX_TILE_KB", "MTGPU_GROUP", "MTGPU_MIN_LDS_KB", "MTGPU_PREFETCH", "MTGPU_ALIGN"):
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                s = m.MotionScanner(params, 0)
            except m.MtgpuError as e:
                print(f"{wl:16s} {name:12s} skipped: {e}")
                continue
            plan = s.plan
            scanners.append((name, s, plan, torch.empty(frames, dtype=torch.uint8, device=dev), []))
        ref = None
        if os.environ.get("AB_NOSYNC") == "1":
            # back-to-back launches, no host sync in between (the bench.py regime)
            for name, s, plan, fl, times in scanners:
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rounds + 2)]
                for e0, e1 in evs:
                    e0.record()
                    (s.check_frames_device_compact if compact else s.check_frames_device)(d_mv, d_off, None, fl)
                    e1.record()
                torch.cuda.synchronize()
                times.extend(e0.elapsed_time(e1) for e0, e1 in evs[2:])
                ref = fl.clone() if ref is None else ref
        for r in range(0 if os.environ.get("AB_NOSYNC") == "1" else rounds + 2):
            for name, s, plan, fl, times in scanners:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                (s.check_frames_device_compact if compact else s.check_frames_device)(d_mv, d_off, None, fl)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times.append(e0.elapsed_time(e1))
                if ref is None:
                    ref = fl.clone()
                assert torch.equal(fl, ref), f"{name} disagrees"
        for name, s, plan, fl, times in scanners:
            t = np.array(times)
            print(f"{wl:16s}{' pan' if pan else ''} {name:12s} fb={plan['counter_bits']:2d} block={plan['block_threads']:4d} "
                  f"bands={plan['bands']} lds={plan['lds_bytes']:6d}  median {np.median(t):.4f} ms  min {t.min():.4f} ms  "
                  f"{alg / np.median(t) / 1e6:7.0f} GB/s (min-time {alg / t.min() / 1e6:7.0f})  motion={int(ref.sum())}")
            s.close()


if __name__ == "__main__":
    main()
