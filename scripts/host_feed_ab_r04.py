#!/usr/bin/env python3
"""Round-4 A/B of the host-fed path (BASELINE config 4 through process_batch on one device): copy-out loop
(MTGPU_PACK / MTGPU_PACK_NT / MTGPU_PACK_PREFETCH) x event wait (MTGPU_EVENT_BLOCKING), interleaved, PASSES passes.
Prints one JSON object (keep it under profiles/).  Needs a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

exe = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "mtgpu_scan_file")
SETTINGS = {
    "scalar (round 3 loop)": {"MTGPU_PACK": "scalar"},
    "auto (vector loop, NT stores)": {},
    "vector loop, ordinary stores": {"MTGPU_PACK_NT": "0"},
    "auto + prefetch 1 KiB": {"MTGPU_PACK_PREFETCH": "1024"},
    "scalar + blocking events": {"MTGPU_PACK": "scalar", "MTGPU_EVENT_BLOCKING": "1"},
    "auto + blocking events": {"MTGPU_EVENT_BLOCKING": "1"},
}
only = os.environ.get("ONLY")
if only:
    SETTINGS = {k: v for k, v in SETTINGS.items() if any(o.strip() in k for o in only.split(","))}
configs = tuple(tuple(int(x) for x in c.split("x")) for c in os.environ.get("CONFIGS", "64x1,16x4").split(","))
out = {}
for p in range(int(os.environ.get("PASSES", "2"))):
    for name, env in SETTINGS.items():
        r = bench.host_fed_batch64(exe, reps=int(os.environ.get("REPS", "400")), extra_env=env, configs=configs)
        keep = {k: {kk: vv for kk, vv in v.items() if kk in ("frames_per_s_wall", "frames_per_s_steady", "wall_ms", "setup_ms",
                                                             "worker_time_share", "cpus_busy", "error")}
                for k, v in r.items() if isinstance(v, dict)}
        out.setdefault(name, []).append(keep)
        print(p, name, {k: (round(v.get("frames_per_s_steady") or 0), round(v.get("frames_per_s_wall") or 0),
                            round(v.get("worker_time_share", {}).get("copy_out_to_pinned", 0), 2),
                            {a: round(b, 1) for a, b in v.get("cpus_busy", {}).items()}) for k, v in keep.items()},
              file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
