#!/usr/bin/env python3
"""BASELINE config 4 through the C++ host layer on one device (bench.host_fed_batch64) under several staging
settings: pinned MiB per batch (MTGPU_BATCH_MB) x staging layout, at 64 streams x 1 worker, 16 x 4 and 4 x 16.
Prints one JSON object; keep it as profiles/r03_host_batch64.json.  Needs a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

exe = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "mtgpu_scan_file")
out = {}
SETTINGS = (("default (4 MiB compact zero-copy batches)", {}),
            ("1 MiB batches", {"MTGPU_BATCH_MB": "1"}),
            ("2 MiB batches", {"MTGPU_BATCH_MB": "2"}),
            ("8 MiB batches", {"MTGPU_BATCH_MB": "8"}),
            ("compact8 with copy commands", {"MTGPU_STAGING": "compact8"}))
if os.environ.get("ONLY_DEFAULT") == "1":
    SETTINGS = SETTINGS[:1]
if os.environ.get("BATCH_AB") == "1":       # 4 vs 8 vs 16 MiB, twice, interleaved
    SETTINGS = tuple((f"{mb} MiB batches, pass {k}", {"MTGPU_BATCH_MB": str(mb)}) for k in (1, 2) for mb in (4, 8, 16))
for name, env in SETTINGS:
    r = bench.host_fed_batch64(exe, reps=int(os.environ.get("REPS", "250")), extra_env=env, runs=0, configs=((64, 1), (16, 4), (4, 16)))
    out[name] = {k: {kk: vv for kk, vv in v.items() if kk in ("frames_per_s_wall", "frames_per_s_steady", "frames_per_s_sum_of_streams", "wall_ms", "wall_ms_until_last_video", "setup_ms",
                                                              "worker_time_share", "held_on_one_device")}
                 for k, v in r.items() if isinstance(v, dict)}
    print(name, {k: (round(v.get("frames_per_s_steady") or 0), round(v["frames_per_s_wall"])) for k, v in out[name].items()},
          file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
