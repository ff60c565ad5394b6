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
SETS = {
    # first call: copy-out loop x event wait, no gate existed yet
    "gate": {
        "scalar (round 3 loop), no gate": {"MTGPU_PACK": "scalar", "MTGPU_CPU_TOKENS": "0"},
        "auto (vector loop, NT stores), no gate": {"MTGPU_CPU_TOKENS": "0"},
        "vector loop, ordinary stores, no gate": {"MTGPU_PACK_NT": "0", "MTGPU_CPU_TOKENS": "0"},
        "auto + prefetch 1 KiB, no gate": {"MTGPU_PACK_PREFETCH": "1024", "MTGPU_CPU_TOKENS": "0"},
        "auto + blocking events, no gate": {"MTGPU_EVENT_BLOCKING": "1", "MTGPU_CPU_TOKENS": "0"},
        "auto, gate = cpu limit": {"MTGPU_CPU_TOKENS": "16"},
        "auto, gate 8": {"MTGPU_CPU_TOKENS": "8"},
        "auto, gate 12": {"MTGPU_CPU_TOKENS": "12"},
        "auto, gate 24": {"MTGPU_CPU_TOKENS": "24"},
        "auto + prefetch 1 KiB, gate = cpu limit": {"MTGPU_PACK_PREFETCH": "1024", "MTGPU_CPU_TOKENS": "16"},
        "scalar, gate = cpu limit": {"MTGPU_PACK": "scalar", "MTGPU_CPU_TOKENS": "16"},
        "auto + blocking events, gate = cpu limit": {"MTGPU_EVENT_BLOCKING": "1", "MTGPU_CPU_TOKENS": "16"},
    },
    # the round's final host layer against each of its parts switched back
    "final": {
        "default (gate 3/4 of the CPU budget, 8 pooled streams, lazy pinning, vector copy-out)": {},
        "a stream per batch (round 3)": {"MTGPU_PIPE_STREAMS": "0"},
        "4 pooled streams": {"MTGPU_PIPE_STREAMS": "4"},
        "16 pooled streams": {"MTGPU_PIPE_STREAMS": "16"},
        "every batch pinned at creation (round 3)": {"MTGPU_PIPE_EAGER": "1"},
        "no CPU gate (round 3)": {"MTGPU_CPU_TOKENS": "0"},
        "scalar copy-out (round 3)": {"MTGPU_PACK": "scalar"},
        "everything as in round 3": {"MTGPU_PIPE_STREAMS": "0", "MTGPU_PIPE_EAGER": "1", "MTGPU_CPU_TOKENS": "0", "MTGPU_PACK": "scalar"},
    },
    # does the 16-MiB batch of round 3 (chosen while the workers were throttled) still pay?
    "batch": {
        "16 MiB batches (default)": {},
        "8 MiB batches": {"MTGPU_BATCH_MB": "8"},
        "4 MiB batches": {"MTGPU_BATCH_MB": "4"},
        "2 MiB batches": {"MTGPU_BATCH_MB": "2"},
    },
    # workers confined to a window of CPUs next to the device (MTGPU_CPU_WINDOW) x CPU tokens
    "window": {
        "no window (workers float over every CPU)": {"MTGPU_CPU_WINDOW": "off"},
        "auto window (1.5 x the CPU budget)": {},
        "window 16": {"MTGPU_CPU_WINDOW": "16"},
        "window 20": {"MTGPU_CPU_WINDOW": "20"},
        "window 28": {"MTGPU_CPU_WINDOW": "28"},
        "window 32": {"MTGPU_CPU_WINDOW": "32"},
        "auto window, 16 tokens": {"MTGPU_CPU_TOKENS": "16"},
        "auto window, 8 tokens": {"MTGPU_CPU_TOKENS": "8"},
        "auto window, no gate": {"MTGPU_CPU_TOKENS": "0"},
        "window 32, 16 tokens": {"MTGPU_CPU_WINDOW": "32", "MTGPU_CPU_TOKENS": "16"},
    },
    "window2": {
        "no window (workers float over every CPU)": {"MTGPU_CPU_WINDOW": "off"},
        "auto window (24 cores next to the device)": {},
        "window 20 cores": {"MTGPU_CPU_WINDOW": "20"},
        "window 32 cores": {"MTGPU_CPU_WINDOW": "32"},
        "window = the far node's cores 0-23": {"MTGPU_CPU_WINDOW": "0-23"},
    },
    "tokens2": {      # CPU tokens once the workers are confined to 20 cores
        "12 tokens (default)": {},
        "10 tokens": {"MTGPU_CPU_TOKENS": "10"},
        "14 tokens": {"MTGPU_CPU_TOKENS": "14"},
        "16 tokens": {"MTGPU_CPU_TOKENS": "16"},
        "no gate": {"MTGPU_CPU_TOKENS": "0"},
    },
    "pack2": {        # the copy-out loop once the feed is bound by the CPU budget (window + gate in force)
        "default (AVX-512BW, NT stores, no software prefetch)": {},
        "prefetch 1 KiB": {"MTGPU_PACK_PREFETCH": "1024"},
        "prefetch 4 KiB": {"MTGPU_PACK_PREFETCH": "4096"},
        "AVX2 loop": {"MTGPU_PACK": "avx2"},
        "AVX2 loop, prefetch 1 KiB": {"MTGPU_PACK": "avx2", "MTGPU_PACK_PREFETCH": "1024"},
        "ordinary stores": {"MTGPU_PACK_NT": "0"},
        "scalar loop": {"MTGPU_PACK": "scalar"},
    },
    "batch2": {
        "16 MiB batches": {"MTGPU_BATCH_MB": "16"},
        "8 MiB batches": {"MTGPU_BATCH_MB": "8"},
        "12 MiB batches": {"MTGPU_BATCH_MB": "12"},
    },
}
SETTINGS = SETS[os.environ.get("SET", "final")]
only = os.environ.get("ONLY")
if only:
    SETTINGS = {k: v for k, v in SETTINGS.items() if any(o.strip() in k for o in only.split(","))}
configs = tuple(tuple(int(x) for x in c.split("x")) for c in os.environ.get("CONFIGS", "64x1,16x4").split(","))
def cpu_stat():
    try:
        return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}


out = {}
for p in range(int(os.environ.get("PASSES", "2"))):
    for name, env in SETTINGS.items():
        c0 = cpu_stat()
        r = bench.host_fed_batch64(exe, reps=int(os.environ.get("REPS", "400")), extra_env=env, runs=0, configs=configs)
        c1 = cpu_stat()
        throttle = {k: c1.get(k, 0) - c0.get(k, 0) for k in ("nr_periods", "nr_throttled", "throttled_usec")}
        keep = {k: {kk: vv for kk, vv in v.items() if kk in ("frames_per_s_wall", "frames_per_s_steady", "wall_ms", "setup_ms",
                                                             "worker_time_share", "cpus_busy", "error",
                                                             "worker_cpu_over_copy_submit_wall", "cpu_gate", "cpu_window")}
                for k, v in r.items() if isinstance(v, dict)}
        keep["cgroup_cpu_stat_delta_incl_file_generation"] = throttle
        out.setdefault(name, []).append(keep)
        print(p, name, {k: (round(v.get("frames_per_s_steady") or 0), round(v.get("frames_per_s_wall") or 0),
                            round(v.get("worker_time_share", {}).get("copy_out_to_pinned", 0), 2),
                            {a: round(b, 1) for a, b in v.get("cpus_busy", {}).items()},
                            round(v.get("worker_cpu_over_copy_submit_wall", 0), 2)) for k, v in keep.items() if "x" in k}, throttle,
              file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
