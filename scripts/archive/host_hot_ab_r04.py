#!/usr/bin/env python3
"""Round-4 probe of the single hot stream (one video, 16 workers, MV bytes cache-resident): what bounds the compact
zero-copy feed at ~45 GB/s when the link reads 55-58 GB/s (profiles/r04_readbw_pinned_host_over_pcie.txt)?
Settings x 2 passes, 120 000 frames per run; frames/s over the window "all workers initialised -> last result".
Needs a GPU."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

exe = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "mtgpu_scan_file")
spec = synth.spec_1080p(seed=9)
n, reps = 12, int(os.environ.get("REPS", "10000"))
spec.events = synth.scripted_events(spec, 60)
frames = [synth.gen_frame(spec, i) for i in range(31, 31 + n)]
SETTINGS = {
    "default": {},
    "eager pinning": {"MTGPU_PIPE_EAGER": "1"},
    "8 MiB batches": {"MTGPU_BATCH_MB": "8"},
    "32 MiB batches": {"MTGPU_BATCH_MB": "32"},
    "64 MiB batches": {"MTGPU_BATCH_MB": "64"},
    "a stream per batch": {"MTGPU_PIPE_STREAMS": "0"},
    "2 pooled streams": {"MTGPU_PIPE_STREAMS": "2"},
    "32 pooled streams": {"MTGPU_PIPE_STREAMS": "32"},
    "no CPU gate": {"MTGPU_CPU_TOKENS": "0"},
    "8 tokens": {"MTGPU_CPU_TOKENS": "8"},
    "copy commands instead of zero-copy": {"MTGPU_STAGING": "compact8"},
    "copy commands, 32 MiB": {"MTGPU_STAGING": "compact8", "MTGPU_BATCH_MB": "32"},
}
threads = int(os.environ.get("THREADS", "16"))
out = {}
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
    path = os.path.join(d, "hot.mtmv")
    m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, [spec.pts_ticks(i) for i in range(n)],
                        frames, key=[1] * n)
    recs = sum(len(f) for f in frames) / n
    for p in range(int(os.environ.get("PASSES", "2"))):
        for name, envx in SETTINGS.items():
            env = dict(os.environ, CHUNK_DURATION_SEC="10", TARGET_FPS="0")
            for k in ("MTGPU_BATCH_MB", "MTGPU_STAGING"):
                env.pop(k, None)
            env.update(envx)
            j = json.loads(subprocess.run([exe, path, "--threads", str(threads), "--repeat", str(reps)], check=True,
                                          capture_output=True, text=True, env=env, timeout=300).stdout)
            w = j["scan_work_us"]
            fps = n * reps / (w * 1e-6)
            rec = {"frames_per_s": fps, "pcie_GBps_of_compact_records": fps * (8 * recs + 10) / 1e9,
                   "copy": j["copy_us"] / threads / w, "submit": j["submit_us"] / threads / w, "wait": j["wait_us"] / threads / w}
            out.setdefault(name, []).append(rec)
            print(p, name, {k: round(v, 3) if v < 10 else round(v) for k, v in rec.items()}, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
