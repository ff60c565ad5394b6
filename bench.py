#!/usr/bin/env python3
"""MV-scan bench (BASELINE.json metric: "MV-scan frames/sec at 1080p grid").

A step = one pass of the hot path over one device-resident batch of synthetic
1080p AVMotionVector arrays (config "Synthetic 1080p MV arrays, 120x68 16px grid,
30 fps stream", dense8x8 = 32 640 records = 1 305 600 B per P-frame; default 16 384 frames = 21 GB
per GPU and step):
    scan kernel (flags per frame)  ->  stream merge kernel (segments per stream)
    [N > 1: one RCCL all_gather of the per-GPU segment lists]
Inputs are resident in HBM before the timed region.  Frames shard across ranks
(weak scaling: every rank owns `--frames` frames of its own streams).

Launching: `python bench.py --gpus N` starts N rank processes itself (one per GPU, RCCL
rendezvous on 127.0.0.1) when it is not already running under a launcher; under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one rank
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline        — the scan kernel: algorithmic bytes / live HIP-event duration vs 8 TB/s
  cpu_baseline    — the C oracle timed on this host's cores on a bounded sample (N=1 only)
  other_workloads — (N=1 only) 4K 240x135 and 4K fine 960x540 scans, same measurement
  host_fed        — (N=1 only) PCIe-inclusive frames/s of the C++ host pipeline (never `value`)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy)

# (workload, params, frames, steps): scanned after the headline at N=1 and reported under
# "other_workloads" (north_star: frames/s on 1080p AND 4K; config 5 = the 960x540 fine grid).
# Every leg scans about 21 GB per launch, like the headline (a launch's ramp and tail are then the same small share
# of every leg; rounds 1-2 used 5 GB for the 16-px grids).
OTHER_WORKLOADS = [("4k_dense8x8", "code_defaults", 4096, 20),
                   # SURVEY.md 8(d) config 3 names both parameter sets: the shipped env (T 4, VECTORS_NEEDED 4)
                   ("4k_dense8x8", "shipped_env", 4096, 20),
                   ("4k_fine", "code_defaults", 1024, 12),
                   # shipped env (VECTORS_NEEDED 4) on the fine grid needs >= 4 records per 4x4 block somewhere to
                   # ever say yes: the dense4 density (4 per block inside moving regions, ragged frames)
                   ("4k_fine_dense4", "shipped_env", 1024, 12),
                   # SURVEY.md 8(d) config 2, the other parameter set and the secondary density
                   # (one record per 16-px cell: 326 KB frames, several per workgroup)
                   ("1080p_dense8x8", "shipped_env", 16384, 20),
                   ("1080p_dense16", "code_defaults", 65536, 20),
                   # SD CCTV stream, one record per 16-px cell: 1200 records = 48 KB per frame, smaller than one
                   # streaming step of a workgroup — eight frames per workgroup, the whole frame in one round trip
                   ("480p_dense16", "code_defaults", 262144, 20)]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=16384,
                    help="frames per GPU per step (16384 dense8x8 1080p frames = 21 GB of records resident in HBM)")
    ap.add_argument("--streams", type=int, default=8, help="streams per GPU (frames split evenly)")
    ap.add_argument("--distinct", type=int, default=60, help="distinct generated frames per GPU (tiled)")
    ap.add_argument("--workload", default="1080p_dense8x8",
                    choices=["1080p_dense8x8", "1080p_dense16", "4k_dense8x8", "4k_dense16", "4k_fine", "4k_fine_dense4", "480p_dense16",
                             "720p_dense16", "480p_dense8x8", "720p_dense8x8"])
    ap.add_argument("--params", default="code_defaults", choices=["code_defaults", "shipped_env"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--no-merge", action="store_true", help="time the scan kernel alone")
    ap.add_argument("--no-others", action="store_true", help="skip the other_workloads leg")
    ap.add_argument("--no-host", action="store_true", help="skip the host-fed (PCIe-inclusive) leg")
    ap.add_argument("--host-runs", type=int, default=3,
                    help="host-fed config-4 shapes: long runs per shape (median / min / max are reported; 0 = the short cold-start run only)")
    ap.add_argument("--host-seconds", type=float, default=21.0,
                    help="... and the least wall time of each long run (creating a worker's pipe and page-locking its staging "
                         "takes ~0.6 s: under 3 %% of 21 s)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 child runs (N = 1 only; the committed figure is replayed)")
    ap.add_argument("--pmc-child", action="store_true",
                    help="internal: the process rocprofv3 profiles for roofline.traffic — builds the workload, launches the "
                         "scan a few times and exits (no JSON line)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N>1 control flow on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--rank-timeout", type=float, default=240.0,
                    help="N > 1 only: a rank whose log and stderr stay silent this many seconds is taken to be hung — "
                         "the launcher terminates every rank and exits 124 (per-rank logs are kept); inside a rank the "
                         "same limit bounds the process-group collectives and arms a traceback dump + exit (0 = off)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the distributed code path (process group, async all_gather of segment lists, "
                         "barriers, max-over-ranks reduction, rank identity gather) even with ONE rank — executes the "
                         "RCCL (`nccl`) branch on a 1-GPU box")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------- launcher

def needs_launch(a, environ):
    """True when this process must start the ranks itself: more than one GPU asked for and no
    launcher (torchrun / the driver) has already made it a rank."""
    return a.gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def rank_environments(n, environ, port):
    """One environment per rank for a single-node run (rendezvous on 127.0.0.1)."""
    envs = []
    for r in range(n):
        e = dict(environ)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        envs.append(e)
    return envs


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_log_dir(environ=None):
    """Where the ranks of an N > 1 run leave their evidence: $MTGPU_BENCH_LOG_DIR if set (the launcher hands it to its
    ranks; tests point it at a temporary directory), else gpurun_out/ when that exists (it is what travels back from a
    GPU box), else beside bench.py."""
    environ = os.environ if environ is None else environ
    if environ.get("MTGPU_BENCH_LOG_DIR"):
        return environ["MTGPU_BENCH_LOG_DIR"]
    d = os.path.join(ROOT, "gpurun_out")
    return d if os.path.isdir(d) else ROOT


def rank_log_path(rank, log_dir=None):
    """bench_rank{r}.err: every rank of an N > 1 run leaves its own evidence (device held, frames scanned, kernel
    time, or the traceback that ended it)."""
    return os.path.join(log_dir or rank_log_dir(), f"bench_rank{rank}.err")


def rank_evidence(rank, log_dir=None):
    """(bytes, newest mtime) over the files a rank writes while it lives: its stage log and its stderr file."""
    size, mtime = 0, 0.0
    for path in (rank_log_path(rank, log_dir), rank_log_path(rank, log_dir) + ".stderr.log"):
        try:
            st = os.stat(path)
        except OSError:
            continue
        size += st.st_size
        mtime = max(mtime, st.st_mtime)
    return size, mtime


def launch_ranks(a, argv, environ=None, popen=subprocess.Popen, clock=time.monotonic, sleep=time.sleep,
                 evidence=None, report=None, log_dir=None):
    """Start a.gpus fresh rank processes of this script (the parent never touches HIP, and no
    process that has is ever re-exec'ed), wait for all of them; rank 0's stdout (the JSON line)
    is ours.  Returns the exit code: non-zero if any rank failed.

    Watchdog (first contact with N > 1 ranks happens on the driver's box): every rank appends a line to
    bench_rank{r}.err at each stage and its stderr goes to bench_rank{r}.err.stderr.log; a live rank whose
    evidence has not grown for --rank-timeout seconds is hung (a collective that never completes, a rendezvous
    that never forms): every rank is terminated (killed 5 s later if it ignores that), the verdict is written to
    bench_launcher.err next to the rank logs, and the exit code is 124.  `clock`, `sleep`, `evidence` and
    `popen` are injectable (tests/test_bench_contract.py); `log_dir` is where the rank logs and bench_launcher.err
    go (default: rank_log_dir()) — the ranks are told through MTGPU_BENCH_LOG_DIR."""
    environ = dict(os.environ if environ is None else environ)
    log_dir = log_dir or rank_log_dir(environ)
    environ["MTGPU_BENCH_LOG_DIR"] = log_dir
    if evidence is None:
        def evidence(r):
            return rank_evidence(r, log_dir)
    envs = rank_environments(a.gpus, environ, free_port())
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    limit = float(getattr(a, "rank_timeout", 0) or 0)
    if report is None:
        def report(text):
            sys.stderr.write(text + "\n")
            try:
                with open(os.path.join(log_dir, "bench_launcher.err"), "a") as f:
                    f.write(text + "\n")
            except OSError:
                pass
    # rank 0 keeps this process's stdout / stderr; the other ranks' stderr is kept in files (a rank that dies
    # takes the run down: the reason must survive it, and the file's growth is a sign of life)
    procs = []
    for r, e in enumerate(envs):
        for stale in (rank_log_path(r, log_dir), rank_log_path(r, log_dir) + ".stderr.log"):
            try:
                os.remove(stale)
            except OSError:
                pass
        if r == 0:
            procs.append(popen(cmd, env=e, stdout=None))
        else:
            with open(rank_log_path(r, log_dir) + ".stderr.log", "wb") as errf:
                procs.append(popen(cmd, env=e, stdout=subprocess.DEVNULL, stderr=errf))
    rc = 0
    pending = {r: p for r, p in enumerate(procs)}
    seen = {r: (evidence(r), clock()) for r in pending}      # rank -> (last evidence, when it last changed)
    killed_at = None
    while pending:
        now = clock()
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                ev = evidence(r)
                if ev != seen[r][0]:
                    seen[r] = (ev, now)
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code
                report(f"bench launcher: rank {r} exited with code {code}; terminating the other ranks "
                       f"(they would wait forever in a collective)")
                for q in pending.values():
                    q.terminate()
                killed_at = now
        if pending and limit > 0 and killed_at is None:
            silent = {r: now - seen[r][1] for r in pending}
            # a rank that has not written anything yet is still importing torch / paging the image in (minutes on a
            # fresh box, longer with N ranks at once): it gets twice the limit before it counts as hung
            hung = [r for r, dt in silent.items() if dt > (limit if seen[r][0][0] > 0 else 2 * limit)]
            if hung:
                report(f"bench launcher: rank(s) {hung} silent for more than {limit:.0f} s (no new line in "
                       f"bench_rank*.err, no stderr) — taken to be hung; terminating all {len(pending)} live ranks. "
                       f"Last stage per rank is the last line of its bench_rank{{r}}.err")
                for q in pending.values():
                    q.terminate()
                killed_at = now
                rc = rc or 124
        if pending and killed_at is not None and now - killed_at > 5.0:
            for q in pending.values():             # ignored SIGTERM (stuck in the driver): SIGKILL
                q.kill()
            killed_at = now + 3600.0               # once
        if pending:
            sleep(0.05)
    return rc


# ---------------------------------------------------------------------------- workloads

def make_spec(workload, seed):
    from mvtrim_amd import synth
    if workload in ("480p_dense16", "480p_dense8x8", "720p_dense16", "720p_dense8x8"):     # SD / 720p streams
        w, h = (640, 480) if workload.startswith("480p") else (1280, 720)
        return synth.StreamSpec(w, h, 16, 2 if workload.endswith("8x8") else 1, seed=seed), (w, h, {})
    if workload == "1080p_dense8x8":
        return synth.spec_1080p(seed=seed, sub=2), (1920, 1080, {})
    if workload == "1080p_dense16":
        return synth.spec_1080p(seed=seed, sub=1), (1920, 1080, {})
    if workload == "4k_dense8x8":
        return synth.spec_4k(seed=seed, sub=2), (3840, 2160, {})
    if workload == "4k_dense16":
        return synth.spec_4k(seed=seed, sub=1), (3840, 2160, {})
    if workload == "4k_fine_dense4":
        return synth.spec_4k_fine_dense(seed=seed), (3840, 2160, dict(block_size=4, block_shift=2))
    return synth.spec_4k_fine(seed=seed), (3840, 2160, dict(block_size=4, block_shift=2))


ARENA_BYTES = 20_800_000_000     # one record arena for every 21 GB leg of a run (see build_workload)


def build_workload(workload, params_name, frames, distinct, seed, dev, arena=None, gop=None):
    """`distinct` generated frames tiled to `frames`, resident on `dev`.  `arena` (uint8 device tensor): the
    records are tiled INTO it instead of into a new allocation — every leg of a bench run then reads the same
    physical memory as the headline (where a 21 GB buffer lands after others were freed moved a leg by 2-5 %)."""
    import torch
    import mvtrim_amd as m
    from mvtrim_amd import synth
    spec, (W, H, gridkw) = make_spec(workload, seed=seed)
    if gop is not None:              # key-frame period (frames without side data); default: the workload's own (30)
        spec.gop = gop
    spec.events = synth.scripted_events(spec, distinct)
    if os.environ.get("AB_PAN") == "1":      # developer scripts only: every MV above the threshold ("camera pan")
        spec.events = [synth.Event(0, distinct, 0, 0, spec.cells_x, spec.cells_y, 9, 3)]
    mv, off, pts, sd = synth.gen_stream(spec, distinct)
    kw = dict(m.config.CODE_DEFAULTS if params_name == "code_defaults" else m.config.SHIPPED_ENV)
    kw.update(gridkw)
    if spec.sub == 1 and params_name == "code_defaults":
        kw["vectors_needed"] = 1      # one record per cell can never collect 2 votes in a cell
    params = m.ScanParams.from_config(W, H, **kw)
    scanner = m.MotionScanner(params, device=dev.index)
    reps = (frames + distinct - 1) // distinct
    counts = np.tile(np.diff(off.astype(np.int64)), reps)[:frames]
    off_big = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    d_tile = torch.from_numpy(mv.view(np.uint8).copy()).to(dev)
    nbytes, tb = int(off_big[-1]) * 40, d_tile.numel()
    if arena is not None and arena.numel() >= nbytes and tb > 0:
        full = nbytes // tb
        arena[: full * tb].view(full, tb)[:] = d_tile                                # broadcast copy, no temporary
        arena[full * tb: nbytes] = d_tile[: nbytes - full * tb]
        d_mv = arena[:nbytes]
    else:
        d_mv = d_tile.repeat(reps)[: int(off_big[-1]) * 40].contiguous()
    del d_tile
    n_records = int(off_big[-1])
    return dict(spec=spec, mv=mv, off=off, params=params, scanner=scanner, reps=reps, d_mv=d_mv,
                d_off=torch.from_numpy(off_big).to(dev), n_records=n_records, frames=frames, distinct=distinct,
                d_flags=torch.empty(frames, dtype=torch.uint8, device=dev),
                alg_bytes=40 * n_records + 9 * frames)   # 40*N_mv + 8 (offset) + 1 (flag) per frame


def time_scan_only(w, steps, warmup=3):
    """One scan call per step, back to back on torch's current stream, timed by the library's own HIP events on that
    stream (mtgpu_profile_enable: before the planning kernels, between them and the scan kernel, after it).  Returns
    the mean SCAN KERNEL ms; w["plan_ms"] = mean ms of the planning kernels ahead of it."""
    import torch
    s = w["scanner"]
    for _ in range(warmup):
        s.check_frames_device(w["d_mv"], w["d_off"], None, w["d_flags"])
    s.profile(True)
    for _ in range(steps):
        s.check_frames_device(w["d_mv"], w["d_off"], None, w["d_flags"])
    pr = s.profile_read()
    s.profile(False)
    torch.cuda.synchronize()
    assert pr["launches"] == steps
    w["plan_ms"] = pr["plan_ms"]
    return pr["scan_ms"]


def time_compact(w, steps, warmup=3):
    """The same batch as 8-byte compact records (what the host dispatcher stages), resident in HBM:
    mtgpu_scan_frames_device_compact.  Returns (mean scan kernel ms, mean planning ms, flags)."""
    import torch
    import mvtrim_amd as m
    s = w["scanner"]
    rec = m.pack_records(w["mv"])
    d_tile = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(w["d_off"].device)
    d_rec = d_tile.repeat(w["reps"])[: w["n_records"] * 8].contiguous()
    del d_tile
    flags = torch.empty(w["frames"], dtype=torch.uint8, device=w["d_off"].device)
    for _ in range(warmup):
        s.check_frames_device_compact(d_rec, w["d_off"], None, flags)
    s.profile(True)
    for _ in range(steps):
        s.check_frames_device_compact(d_rec, w["d_off"], None, flags)
    pr = s.profile_read()
    s.profile(False)
    torch.cuda.synchronize()
    return pr["scan_ms"], pr["plan_ms"], flags.cpu().numpy()


def host_fed_leg(spec, mv, off, runs=3, seconds=21.0):
    """PCIe-inclusive rate of the host dispatcher (never `value`): the C++ front end mtgpu_scan_file
    fed by 16 worker threads from a 12-frame stream of the headline workload presented many times (MV
    bytes cache-resident, as when a decoder thread has just written them), with the default
    staging (8-byte compact records, zero-copy) and with round 1's path (40-byte records, H2D
    copies): three runs each (separate processes, ~2 s of scanning), median / min / max.  Then BASELINE
    config 4 (host_fed_batch64): a cold start and `runs` runs of >= `seconds` per shape."""
    import tempfile
    import mvtrim_amd as m
    exe = os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "mtgpu_scan_file")
    if not os.path.exists(exe):
        return {"error": "mtgpu_scan_file not built"}
    # Run lengths (round 4): the rate is measured from "all workers initialised" to the last result, and a pipe
    # page-locks its 2nd and 3rd batch on first use, i.e. INSIDE that window (16 workers x 2 x 16 MiB = 0.1 s at the
    # driver's ~5 GB/s): rounds 2-3 ran 6000 frames (35 ms of work), which now measures the page-locking, not the feed
    n, workers = 12, min(16, len(os.sched_getaffinity(0)))
    reps_of = {"compact8_zero_copy": 30000, "aos40_copy": 6000}          # about 2 s of scanning per run each
    frames = [mv[int(off[i]):int(off[i + 1])] for i in range(1, 1 + n)]       # P-frames 1..12 of the tile
    out = {"source": f"{n}-frame 1080p dense8x8 stream repeated {reps_of['compact8_zero_copy']}x / {reps_of['aos40_copy']}x "
                     f"({n * reps_of['compact8_zero_copy']} / {n * reps_of['aos40_copy']} frames), cache-resident",
           "workers": workers, "gpus": 1, "front_end": "mtgpu_scan_file (C++ host layer: chunks -> pinned pipe -> scan -> merge)"}
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=d) as tmp:
        path = os.path.join(tmp, "hot.mtmv")
        m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps,
                            [spec.pts_ticks(i) for i in range(n)], frames, key=[1] * n)
        for name, staging in (("compact8_zero_copy", "compact8_zc"), ("aos40_copy", "aos40")):
            env = dict(os.environ, CHUNK_DURATION_SEC="10", TARGET_FPS="0", MTGPU_STAGING=staging)
            env.pop("MTGPU_BATCH_MB", None)
            # one GPU only (the host layer would spread its workers over every visible device)
            env["HIP_VISIBLE_DEVICES"] = (os.environ.get("HIP_VISIBLE_DEVICES") or "0").split(",")[0]
            reps = reps_of[name]
            rates = []
            for _ in range(3):
                r = subprocess.run([exe, path, "--threads", str(workers), "--repeat", str(reps)], capture_output=True,
                                   text=True, env=env, timeout=120)
                if r.returncode != 0:
                    return dict(out, error=(r.stderr or "mtgpu_scan_file failed")[-300:])
                j = json.loads(r.stdout)
                rates.append(n * reps / max(j["scan_work_us"] * 1e-6, 1e-9))
            rates.sort()
            out[name + "_frames_per_s"] = rates[1]                    # the median of three runs
            out[name + "_spread"] = {"median": rates[1], "min": rates[0], "max": rates[2], "runs": 3}
    out["speedup"] = out["compact8_zero_copy_frames_per_s"] / out["aos40_copy_frames_per_s"]
    # bytes that cross PCIe per frame: the staged records + offset + side-data byte in, one flag byte out
    recs = float(np.mean([len(f) for f in frames]))
    out["pcie_GBps"] = {"compact8_zero_copy": out["compact8_zero_copy_frames_per_s"] * (8 * recs + 10) / 1e9,
                        "aos40_copy": out["aos40_copy_frames_per_s"] * (40 * recs + 10) / 1e9,
                        "note": "PCIe gen5 x16: 64 GB/s raw, about 55 GB/s achievable one way"}
    out["note"] = ("PCIe-inclusive: what a real decode pipeline gets per GPU; the link, not the kernel, is the limit "
                   "(the resident `value` is ~30x higher)")
    try:
        out["config4_64_streams"] = host_fed_batch64(exe, runs=runs, seconds=seconds)
    except Exception as e:      # informational
        out["config4_64_streams"] = {"error": repr(e)}
    return out


def host_fed_batch64(exe, n=12, reps=600, extra_env=None, configs=((64, 1), (16, 4)), runs=3, seconds=21.0):
    """BASELINE config 4 through the product-shaped path on ONE device: 64 distinct-seed 1080p dense8x8 streams
    (12 distinct frames each) through process_batch of the C++ host layer at 64 streams x 1 worker and 16 streams x 4
    workers; default staging (compact, zero-copy).  Per shape: one short run (`reps` presentations of every stream —
    a few seconds, in which creating the pipes and page-locking their staging is > 10 % of the wall: kept as
    `cold_start`) and `runs` long ones of at least `seconds` each (21 s: set-up < 3 % of the wall), reported as
    median / min / max.  The top-level figures of a shape are the medians of the long runs."""
    import tempfile
    import mvtrim_amd as m
    from mvtrim_amd import synth
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    res = {"source": f"64 distinct-seed {n}-frame 1080p dense8x8 streams, page-cache resident; per shape a cold start "
                     f"({reps} presentations of every stream) and {runs} runs of >= {seconds:.0f} s", "gpus": 1}
    with tempfile.TemporaryDirectory(dir=d) as tmp:
        paths = []
        for k in range(64):
            spec = synth.spec_1080p(seed=2000 + k, sub=2)
            spec.events = [synth.Event(1, 1 + n // 2, 10 + k, 12 + k % 40, 4, 3, 9, 2)]
            frames = [synth.gen_frame(spec, i) for i in range(1, 1 + n)]
            path = os.path.join(tmp, f"cam{k:02d}.mtmv")
            m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps,
                                [spec.pts_ticks(i) for i in range(n)], frames, key=[1] * n)
            paths.append(path)
        env = dict(os.environ, CHUNK_DURATION_SEC="10", TARGET_FPS="0")
        for k in ("MTGPU_BATCH_MB", "MTGPU_STAGING"):
            env.pop(k, None)
        env["HIP_VISIBLE_DEVICES"] = (os.environ.get("HIP_VISIBLE_DEVICES") or "0").split(",")[0]
        env.update(extra_env or {})

        def one(streams, threads, reps_):
            r = subprocess.run([exe] + paths + ["--streams", str(streams), "--threads", str(threads), "--repeat",
                                                str(reps_), "--summary", "--outdir", tmp], capture_output=True,
                               text=True, env=env, timeout=600)
            if r.returncode != 0:
                return {"error": (r.stderr or "mtgpu_scan_file failed")[-300:]}
            lines = [json.loads(ln) for ln in r.stdout.strip().splitlines()]
            s = [j["batch_summary"] for j in lines if "batch_summary" in j][0]
            jobs = [j for j in lines if "batch_summary" not in j]
            workers = streams * threads
            wall = max(s["wall_us"], 1) * 1e-6
            busy = max(s["decode_us"] + s["analyze_us"], 1)
            setup_ms = {"mtgpu_create_total": s["held"].get("ctx_create_us", 0) / 1e3,
                        "mtgpu_pipe_create_per_worker": s["held"].get("pipe_create_us", 0) / 1e3 / max(workers, 1),
                        # page-locking incl. the batches pinned later, on first use (summed over the run / per worker)
                        "page_locking_per_worker": s["held"].get("pin_us", 0) / 1e3 / max(workers, 1),
                        "batches_pinned": s["held"].get("pinned_batches", 0)}
            return {
                "streams": streams, "workers_per_stream": threads, "repeat": reps_, "frames": s["frames_scanned"], "jobs": s["jobs"],
                "frames_per_s_wall": s["frames_scanned"] / wall,          # includes creating the contexts + pinned pipes
                # steady state: all frames over the window [first worker of any video ready, last result of any
                # video] — pipes persist from video to video, so set-up happens before it and tear-down after it
                "frames_per_s_steady": s["frames_scanned"] / max(s.get("scan_window_us", 0) * 1e-6, 1e-9)
                                       if s.get("scan_window_us", 0) > 0 else None,
                # (rounds' own rates added up: an upper estimate, kept for comparison with earlier figures)
                "frames_per_s_sum_of_streams": (float(np.mean([j["frames_scanned"] / max(j["scan_work_us"] * 1e-6, 1e-9)
                                                               for j in jobs])) * streams) if jobs else None,
                "wall_ms": s["wall_us"] / 1e3, "wall_ms_until_last_video": s.get("scan_wall_us", 0) / 1e3,
                # share of the wall a worker spent creating its pipe and page-locking its staging
                "setup_share_of_wall": (setup_ms["mtgpu_pipe_create_per_worker"] + setup_ms["page_locking_per_worker"]) / (wall * 1e3),
                # CPU time of the whole process over the run (getrusage) / wall: how many CPUs it kept busy
                "cpus_busy": {"user": s.get("cpu_user_us", 0) / max(s["wall_us"], 1), "sys": s.get("cpu_sys_us", 0) / max(s["wall_us"], 1)},
                # CPU time the worker threads got / the wall time they spent copying + submitting (waiting sleeps):
                # well below 1 = workers were runnable but not running (more runnable threads than the CPU budget)
                "worker_cpu_over_copy_submit_wall": s.get("worker_cpu_us", 0) / max(s["copy_us"] + s["submit_us"], 1),
                "cpu_gate": {"tokens": s.get("gate_tokens", 0), "wait_share_of_worker_time": s.get("gate_wait_us", 0) / (workers * wall * 1e6)},
                # CPUs next to the device that the worker threads are confined to (0 = not pinned: no CPU quota in force)
                "cpu_window": {"cpus": s.get("cpu_window", 0), "first": s.get("cpu_window_first", -1)},
                "setup_ms": setup_ms,
                "worker_time_share": {"init": s["init_us"] / (workers * wall * 1e6),
                                      "reading_frames": s["decode_us"] / busy,
                                      "copy_out_to_pinned": s["copy_us"] / busy,
                                      "submit_calls": s["submit_us"] / busy,
                                      "waiting_for_gpu": s["wait_us"] / busy},
                "held_on_one_device": {"contexts": s["held"]["contexts"], "pipes": s["held"]["pipes"],
                                       "hip_streams": s["held"]["hip_streams"], "hip_events": s["held"]["hip_events"],
                                       "mem_pools": s["held"]["mem_pools"],
                                       "pinned_MiB": s["held"]["pinned_bytes"] / 2**20,
                                       "device_MiB": s["held"]["device_bytes"] / 2**20}}

        for streams, threads in configs:
            cold = one(streams, threads, reps)
            if "error" in cold or runs <= 0:
                res[f"{streams}x{threads}"] = cold if "error" in cold else dict(cold, cold_start=None, long_runs=0)
                continue
            # long runs: enough presentations for >= `seconds` of wall at the cold run's steady rate
            rate = cold["frames_per_s_steady"] or cold["frames_per_s_wall"]
            # (a warm run is ~10 % faster than the cold one the rate comes from)
            reps_long = max(reps, int(np.ceil(1.12 * seconds * rate / (64 * n))))
            longs = [one(streams, threads, reps_long) for _ in range(runs)]
            bad = [x for x in longs if "error" in x]
            if bad:
                res[f"{streams}x{threads}"] = dict(bad[0], cold_start=cold)
                continue
            by_wall = sorted(longs, key=lambda x: x["frames_per_s_wall"])
            mid = dict(by_wall[len(by_wall) // 2])                 # the median run, whole — its breakdown belongs to its rate
            for key in ("frames_per_s_wall", "frames_per_s_steady"):
                vals = sorted(x[key] for x in longs if x[key])
                mid[key + "_spread"] = {"median": vals[len(vals) // 2], "min": vals[0], "max": vals[-1], "runs": len(vals)}
                mid[key] = vals[len(vals) // 2]
            mid["cold_start"] = {k_: cold[k_] for k_ in ("repeat", "frames", "frames_per_s_wall", "frames_per_s_steady", "wall_ms",
                                                        "setup_share_of_wall", "setup_ms")}
            res[f"{streams}x{threads}"] = mid
    return res


def replayed_traffic(workload, params_name, frames):
    """(HBM bytes per launch, where the number comes from) for a bench leg, from the committed summary of the
    builder's own rocprofv3 --pmc passes (scripts/profile_round.sh) — replayed, never measured by this run."""
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(tp)).get(f"{workload}:{params_name}:{frames}")
    except Exception:
        rec = None
    if not rec:
        return None, None
    return rec["hbm_bytes_per_launch"], (
        f"replayed, not measured in this run: profiles/pmc_traffic.json [{workload}:{params_name}:{frames}], "
        f"builder-run rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (gfx950 x2 FETCH correction), "
        f"{rec.get('collected') or 'round 2'}")


def load_calib():
    """libmtgpu_calib.so (csrc/calib/): read-only kernels in the scan's load shapes, for the measured read ceiling.
    A library of its own — the product (libmtgpu.so) carries no calibration code.  Returns a callable
    (device, ptr, bytes, shape, chunk, lds_bytes, idle_every, stream) that raises on error."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "motion-estimated-video-trimmer_amd", "libmtgpu_calib.so"))
    fn = lib.mtcalib_read_ceiling
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.mtcalib_last_error.restype = C.c_char_p

    def call(device, ptr, nbytes, shape, chunk, lds_bytes, idle_every, stream):
        if fn(device, ptr, nbytes, shape, chunk, lds_bytes, idle_every, stream) != 0:
            raise RuntimeError("mtcalib_read_ceiling: " + lib.mtcalib_last_error().decode())
    return call


PMC_KERNELS = ("scan_frames_kernel", "plan_scatter_kernel", "plan_count_kernel")     # one scan call = these launches


def parse_pmc_dir(d, counter):
    """(sum of `counter` over the library's kernels per scan launch, scan launches seen) from the
    *_counter_collection.csv rocprofv3 wrote under `d`.  The planning kernels' bytes (offsets in, work list out) are
    added to the scan kernel's: `traffic` is what one scan CALL moves."""
    import csv
    import glob
    hits = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True))
    if not hits:
        raise RuntimeError(f"no *_counter_collection.csv under {d}")
    total, launches = 0.0, 0
    for path in hits:
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                if any(k in name for k in PMC_KERNELS):
                    total += float(row["Counter_Value"])
                    launches += "scan_frames_kernel" in name
    if launches == 0:
        raise RuntimeError(f"{counter}: no scan_frames_kernel dispatch in {d}")
    return total / launches, launches


def pmc_bytes(fetch_kb, write_kb):
    """HBM bytes from the two counters as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: unit KB, FETCH_SIZE
    reports half the bytes of wide coalesced streaming reads (x2), WRITE_SIZE is exact."""
    return fetch_kb * 1024.0 * 2.0 + write_kb * 1024.0


def measure_traffic(a, argv, which=None, run=subprocess.run, tmp_root=None):
    """roofline.traffic measured IN this run (N = 1): before this process touches the GPU, two fresh child processes
    of this script run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes,
    no other trace domain, the interpreter directly after `--`: no env / shell / launcher hop), each building the
    headline workload and launching the scan four times.  Returns (bytes per launch, source text, detail dict), or
    (None, reason, None) when rocprofv3 is missing or a child fails — the caller then replays the committed figure."""
    import shutil
    import tempfile
    which = which or shutil.which
    exe = which("rocprofv3") or (os.path.exists("/opt/rocm/bin/rocprofv3") and "/opt/rocm/bin/rocprofv3")
    if not exe:
        return None, "rocprofv3 not found", None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", a.workload, "--params", a.params,
             "--frames", str(a.frames), "--distinct", str(a.distinct)]
    out = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir=tmp_root or "/tmp") as tmp:
        env = dict(os.environ, TMPDIR=tmp)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-f", "csv", "-d", d, "--"] + child
            try:
                r = run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=150)
            except Exception as e:
                return None, f"rocprofv3 --pmc {counter} child did not finish: {e!r}", None
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} child exited {r.returncode}: {(r.stderr or '')[-200:]}", None
            try:
                out[counter] = parse_pmc_dir(d, counter)
            except Exception as e:
                return None, f"rocprofv3 --pmc {counter}: {e}", None
    (f_kb, nf), (w_kb, nw) = out["FETCH_SIZE"], out["WRITE_SIZE"]
    detail = {"FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB_raw": w_kb, "launches_averaged": [nf, nw],
              "kernels_summed": list(PMC_KERNELS), "seconds": round(time.perf_counter() - t0, 1)}
    return pmc_bytes(f_kb, w_kb), (
        "measured in this run: two child processes of bench.py under rocprofv3 --kernel-trace --pmc FETCH_SIZE and "
        "--pmc WRITE_SIZE (separate passes; gfx950: counter unit KB, FETCH_SIZE x2), mean over "
        f"{nf} scan launches of the headline workload, planning kernels included"), detail


def pmc_child(a):
    """What rocprofv3 profiles for measure_traffic: the headline workload resident in HBM, four scan launches."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    w = build_workload(a.workload, a.params, a.frames, a.distinct, 1000, dev)
    for _ in range(4):
        w["scanner"].check_frames_device(w["d_mv"], w["d_off"], None, w["d_flags"])
    torch.cuda.synchronize()
    w["scanner"].close()


def roofline_of(alg_bytes, kern_ms):
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "scan_frames_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "kernel_ms": kern_ms,
            "algorithmic_bytes_per_launch": alg_bytes}


def other_workloads(dev, distinct, arena=None):
    """4K / fine-grid scans after the headline (N=1): frames/s, kernel ms, roofline fraction; every
    batch's flags are checked to be the tile's flags repeated and the tile's flags == the oracle's."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob     # checker only
    out = []
    for (wl, pn, frames, steps) in OTHER_WORKLOADS:
        dd = min(distinct, 60)      # the scripted events start at second 1 (frame 30): a 60-frame tile holds one
        w = build_workload(wl, pn, frames, dd, 1000, dev, arena)
        kern_ms = time_scan_only(w, steps)
        flags = w["d_flags"].cpu().numpy()
        tile = flags[:dd]
        assert np.array_equal(flags, np.tile(tile, w["reps"])[:frames]), f"{wl}: flags are not tile-periodic"
        want = ob.scan_frames(w["params"], w["mv"], w["off"], None, nthreads=min(len(os.sched_getaffinity(0)), 16))
        assert np.array_equal(tile, want), f"{wl}: GPU flags differ from the oracle"
        # a leg whose answer is constant times nothing but the stream: every leg must say yes AND no
        assert 0 < int(flags.sum()) < frames, f"{wl}/{pn}: degenerate workload ({int(flags.sum())} motion frames of {frames})"
        r = roofline_of(w["alg_bytes"], kern_ms)
        p = w["params"]
        out.append({"workload": f"synthetic {wl} MV arrays, {p.grid_w}x{p.grid_h} grid, {frames} frames "
                                f"({dd} distinct tiled), params={pn}"
                                + (" with VECTORS_NEEDED 1 (one record per cell)" if p.vectors_needed == 1 and pn == "code_defaults" else ""),
                    "frames_per_s": frames / ((kern_ms + w["plan_ms"]) * 1e-3), "kernel_ms": kern_ms, "plan_ms": w["plan_ms"],
                    "steps": steps, "achieved_GBps": r["achieved"], "frac": r["frac"],
                    "frac_of_call": r["achieved"] * kern_ms / (kern_ms + w["plan_ms"]) / HBM_PEAK_GBS, "plan": w["scanner"].plan,
                    "algorithmic_bytes_per_launch": w["alg_bytes"],
                    "traffic": replayed_traffic(wl, pn, frames)[0],
                    "traffic_source": replayed_traffic(wl, pn, frames)[1],
                    "motion_frames_in_batch": int(flags.sum())})
        w["scanner"].close()
        del w
        torch.cuda.empty_cache()
    # the headline workload once more as 8-byte compact records resident in HBM (the layout the
    # host dispatcher stages): 5x fewer bytes per frame, so frames/s rise; its own byte count is used
    for cframes in (4096, 16384):
        try:
            w = build_workload("1080p_dense8x8", "code_defaults", cframes, min(distinct, 60), 1000, dev, arena)
            ref_ms = time_scan_only(w, 10)
            k8, p8, f8 = time_compact(w, 40)
            assert np.array_equal(f8, w["d_flags"].cpu().numpy()), "compact flags differ from the 40-byte scan"
            cbytes = 8 * w["n_records"] + 9 * w["frames"]
            out.append({"workload": f"synthetic 1080p_dense8x8 as COMPACT 8-byte records (src/dst int16 x4), {cframes} frames",
                        "frames_per_s": w["frames"] / (k8 * 1e-3), "kernel_ms": k8, "steps": 40,
                        "achieved_GBps": cbytes / (k8 * 1e-3) / 1e9, "frac": cbytes / (k8 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "plan_ms": p8, "frac_of_call": cbytes / ((k8 + p8) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "bytes_per_launch": cbytes, "speedup_vs_40_byte_records": ref_ms / k8})
            w["scanner"].close()
            del w
            torch.cuda.empty_cache()
        except Exception as e:      # informational leg: never take the headline down
            out.append({"workload": f"1080p compact records, {cframes} frames", "error": repr(e)})
    # key frames: a frame without side data never gets a workgroup (the work list, scan_kernels.hip plan_*), so the rate
    # must not depend on where such frames fall.  The headline workload's frames with a key frame every 8 frames
    # (round 5, one workgroup per frame: one of the 8 XCDs without work, -4.5 .. -8 %) and with none at all.
    try:
        rates = {}
        for name, gop in (("no_key_frames", 0), ("key_frame_every_8", 8), ("no_key_frames_again", 0), ("key_frame_every_8_again", 8)):
            w = build_workload("1080p_dense8x8", "code_defaults", 4096 if gop == 0 else 4680, 64, 1000, dev, arena, gop=gop)
            ms = time_scan_only(w, 10) + w["plan_ms"]           # the whole call: planning kernels + scan kernel
            fl = w["d_flags"].cpu().numpy()
            assert np.array_equal(fl, np.tile(fl[:64], w["reps"])[: w["frames"]]), "flags are not tile-periodic"
            rates[name] = w["alg_bytes"] / (ms * 1e-3) / 1e9
            w["scanner"].close()
            del w
            torch.cuda.empty_cache()
        out.append({"workload": "1080p_dense8x8, 4096 frames with records: no key frames vs a key frame (no side data) "
                                "every 8 frames (4680 frames)", "achieved_GBps": rates,
                    "frac": min(rates.values()) / HBM_PEAK_GBS,
                    "key_frames_every_8_over_none": (rates["key_frame_every_8"] + rates["key_frame_every_8_again"]) /
                    (rates["no_key_frames"] + rates["no_key_frames_again"])})
    except AssertionError:
        raise
    except Exception as e:          # informational leg
        out.append({"workload": "key-frame period A/B (every 8 frames vs none)", "error": repr(e)})
    return out


# ---------------------------------------------------------------------------- one rank

def device_identity(torch, dev, rank, local):
    """Which physical device this rank holds, for the driver to check (N ranks must show N distinct bus ids)."""
    p = torch.cuda.get_device_properties(dev)
    ident = {"rank": rank, "local_rank": local, "device_index": dev.index, "name": p.name, "pid": os.getpid(),
             "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES")),
             "device_count": torch.cuda.device_count()}
    try:
        ident["pci_bus_id"] = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except AttributeError:
        ident["pci_bus_id"] = None
    try:
        ident["uuid"] = str(p.uuid)
    except AttributeError:
        ident["uuid"] = None
    return ident


def prepare_rank_environment(environ, multi):
    """What a rank of a multi-process run needs in its environment before the GPU runtime starts.  dmabuf IPC only on
    this pool: RCCL's first exchange between two processes fails under the legacy IPC mode ("hipIpcGetMemHandle:
    invalid argument").  Our own launcher sets it per rank (rank_environments); under `python -m torch.distributed.run`
    — the driver's command for N > 1 — nobody else does.  A plain environment write; nothing is re-exec'ed."""
    if multi:
        environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return environ


def run_rank(a):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not a.force_dist:
        return _run_rank(a)
    try:
        return _run_rank(a)
    except BaseException:
        import traceback
        with open(rank_log_path(rank), "a") as f:
            f.write(f"rank {rank} of {world} FAILED\n" + traceback.format_exc())
        raise


def _run_rank(a):
    # stdout carries exactly ONE line, the JSON: anything libraries print on fd 1 (gloo's connection
    # notes, RCCL info lines) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or a.force_dist          # the distributed code path (normally: more than one rank)
    prepare_rank_environment(os.environ, multi)       # before `import torch`: before anything touches the GPU
    if a.force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist
    import mvtrim_amd as m
    from mvtrim_amd import dist as mdist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if a.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ident = device_identity(torch, dev, rank, local)
    hang = {"limit": 0.0, "file": None}

    def stage(name, **kw):
        """One line per stage in this rank's log: evidence for the driver, and the launcher's sign of life.  Every
        stage also re-arms the hang dump: it measures silence since the last sign of life (like the launcher's
        watchdog), not the lifetime of a healthy run."""
        if multi:
            with open(rank_log_path(rank), "a") as f:
                f.write(json.dumps({"stage": name, "t": round(time.time(), 3), **kw}) + "\n")
            if hang["file"] is not None:
                import faulthandler
                faulthandler.cancel_dump_traceback_later()
                faulthandler.dump_traceback_later(2 * hang["limit"], exit=True, file=hang["file"])

    if multi:
        with open(rank_log_path(rank), "w") as f:
            f.write(json.dumps({"stage": "start", **ident}) + "\n")
        limit = float(a.rank_timeout or 0)
        if limit > 0 and world > 1:
            # a rank that is still here after 2 x the limit dumps every thread's stack into its log and exits
            # (under torchrun there is no launcher of ours to notice a hang; the driver's kill would leave nothing)
            import faulthandler
            hang["limit"], hang["file"] = limit, open(rank_log_path(rank) + ".hang.log", "w")
            faulthandler.dump_traceback_later(2 * limit, exit=True, file=hang["file"])
        import datetime
        pg_kw = {"timeout": datetime.timedelta(seconds=limit)} if limit > 0 else {}
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **pg_kw)
        else:
            dist.init_process_group("gloo", **pg_kw)
        stage("process_group_ready", backend=a.backend, world=world)

    # ---------------- synthetic input: `distinct` generated frames, tiled to `frames`
    arena = None
    if world == 1 and not a.no_others:
        try:                        # first allocation of the process: the headline and every 21 GB leg live in it
            arena = torch.empty(ARENA_BYTES, dtype=torch.uint8, device=dev)
        except Exception:
            arena = None
    w = build_workload(a.workload, a.params, a.frames, a.distinct, 1000 + rank, dev, arena)
    spec, mv, off, params, scanner = w["spec"], w["mv"], w["off"], w["params"], w["scanner"]
    d_mv, d_off, d_flags, n_records, alg_bytes, reps = (w["d_mv"], w["d_off"], w["d_flags"], w["n_records"],
                                                         w["alg_bytes"], w["reps"])

    # streams: frames split evenly; pts restart per stream at 30 fps
    S = max(1, min(a.streams, a.frames))
    per = a.frames // S
    stream_off = np.array([i * per for i in range(S)] + [a.frames], dtype=np.int64)
    pts_big = np.concatenate([np.array([spec.pts_seconds(i) for i in range(stream_off[s + 1] - stream_off[s])])
                              for s in range(S)])
    mp = np.concatenate([m.MergeParams(duration=float(stream_off[s + 1] - stream_off[s]) / spec.fps).to_record()
                         for s in range(S)])
    d_pts = torch.from_numpy(pts_big).to(dev)
    d_soff = torch.from_numpy(stream_off).to(dev)
    d_mp = torch.from_numpy(mp.view(np.uint8).copy()).to(dev)
    SEG_CAP = 64

    # two sets of merge outputs / gather buffers: the gather of step i overlaps the scan of step i+1
    packed_w = SEG_CAP * 16 + m.MERGE_RESULT_DTYPE.itemsize
    outs = [(torch.zeros((S, SEG_CAP, 2), dtype=torch.float64, device=dev),
             torch.zeros((S, m.MERGE_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev),
             torch.empty(2 * a.frames, dtype=torch.float64, device=dev)) for _ in range(2)]
    gath = [torch.empty((world, S, packed_w), dtype=torch.uint8, device=dev if a.backend == "nccl" else "cpu")
            for _ in range(2)] if multi else None
    pending = [None, None]
    counter = [0]
    # Two HIP streams, two sets of buffers: the stream-merge kernel (and, N > 1, the gather) of step i runs on its
    # own stream while the scan of step i+1 already streams records on the launch stream — the merge is 8
    # workgroups for ~6 us, there is no reason to hold 256 CUs back for it.  Events order the hand-offs:
    # scan(k) -> merge(k) (flags[k] complete), merge(k) -> scan(k) two steps later (flags[k] / outputs[k] free).
    scan_stream = torch.cuda.current_stream(dev)
    merge_stream = torch.cuda.Stream(device=dev)
    flag_bufs = [d_flags, torch.empty_like(d_flags)]
    scan_done = [torch.cuda.Event() for _ in range(2)]
    merge_done = [None, None]

    def step(i=None):
        k = counter[0] & 1
        counter[0] += 1
        if merge_done[k] is not None:
            scan_stream.wait_event(merge_done[k])
        # (timed by the library's own event triple on this stream: mtgpu_profile_enable below)
        scanner.check_frames_device(d_mv, d_off, None, flag_bufs[k])
        if a.no_merge:
            return None
        scan_done[k].record(scan_stream)
        with torch.cuda.stream(merge_stream):
            merge_stream.wait_event(scan_done[k])
            if pending[k] is not None:          # buffer set k is still being gathered (two steps ago)
                pending[k].wait()
                pending[k] = None
            seg, res = scanner.merge_streams_device(flag_bufs[k], d_pts, d_soff, d_mp, True, SEG_CAP, out=outs[k])
            if multi:
                # the only exchange step of the path: per-GPU segment lists to every rank (RCCL over xGMI),
                # asynchronous so that it overlaps the following scans
                packed = mdist.pack_segment_lists(seg, res)
                if a.backend == "nccl":
                    pending[k] = dist.all_gather_into_tensor(gath[k], packed, async_op=True)
                else:       # rehearsal only: gloo moves the lists through host memory
                    mdist.gather_segment_lists(seg.cpu(), res.cpu(), out=gath[k])
            merge_done[k] = torch.cuda.Event()
            merge_done[k].record(merge_stream)
        return seg, res

    def finish():
        with torch.cuda.stream(merge_stream):
            for k in (0, 1):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None
        merge_stream.synchronize()

    stage("workload_resident", frames=a.frames)
    for _ in range(a.warmup):
        step()
    finish()
    torch.cuda.synchronize()
    # HIP events on the launch stream, recorded by the library around its kernels: before the planning kernels,
    # between them and the scan kernel, after the scan kernel — one more marker per step than an event pair around the call
    if os.environ.get("MTGPU_BENCH_NO_PROFILE") != "1":       # (developer A/B: what the three event markers per step cost)
        scanner.profile(True)
    stage("warm")
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(a.steps):
        out = step(i)
    finish()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_local = dt
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    prof = scanner.profile_read()
    scanner.profile(False)
    if os.environ.get("MTGPU_BENCH_NO_PROFILE") == "1":
        prof = {"launches": a.steps, "scan_ms": dt_local / a.steps * 1e3, "plan_ms": 0.0}
    assert prof["launches"] == a.steps, prof
    kern_ms, plan_ms = prof["scan_ms"], prof["plan_ms"]
    # calibration (outside the timed region): kernels that ONLY read the same record buffer — both load shapes, three
    # chunk sizes, 20 launches back to back each (the regime of the timed loop); the ceiling is the best of them
    calib = load_calib()
    st = torch.cuda.current_stream(dev).cuda_stream
    nbytes = (d_mv.numel() // 16) * 16
    frame_bytes = 40 * int(spec.records_per_frame)                                # a P-frame of this workload (ragged workloads: its nominal size)
    sweep = {}
    configs = []
    for shape, sname, chunks, idle in ((2, "12of40+arith", (frame_bytes, 1280 * 1024), 30), (2, "12of40+arith", (frame_bytes, 1280 * 1024), 0),
                                       (3, "12of40+arith+lds", (frame_bytes,), 30), (1, "12of40", (1280 * 1024, 5 * 1024 * 1024), 0),
                                       (0, "16B", (1280 * 1024, 5 * 1024 * 1024), 0)):
        configs += [(f"{sname}/{chunk}" + (f"/idle{idle}" if idle else ""), shape, chunk, idle) for chunk in chunks
                    if nbytes >= 256 * chunk]        # (a test-sized batch calibrates nothing: at least a workgroup per CU)
    times = {name: [] for name, _, _, _ in configs}
    for order in (configs, configs[::-1]):       # two passes in opposite orders: no configuration owes its figure to its place
        for name, shape, chunk, idle in order:
            for _ in range(2):
                calib(dev.index, d_mv.data_ptr(), nbytes, shape, chunk, scanner.plan["lds_bytes"], idle, st)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for c0, c1 in evs:           # an event pair around every launch, exactly as the scan kernel is timed
                c0.record()
                calib(dev.index, d_mv.data_ptr(), nbytes, shape, chunk, scanner.plan["lds_bytes"], idle, st)
                c1.record()
            torch.cuda.synchronize()
            times[name] += [c0.elapsed_time(c1) for c0, c1 in evs]
    for name, t in times.items():
        sweep[name] = nbytes / (max(float(np.mean(t)), 1e-6) * 1e-3) / 1e9       # (a buffer smaller than one chunk launches nothing)
    # ... and the scan itself once more, timed the same way in the same phase of the run (clocks and the memory
    # system's state drift over a run: the timed loop ran seconds earlier)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for c0, c1 in evs:
        c0.record()
        scanner.check_frames_device(d_mv, d_off, None, flag_bufs[0])
        c1.record()
    torch.cuda.synchronize()
    scan_in_sweep = alg_bytes / (max(float(np.mean([c0.elapsed_time(c1) for c0, c1 in evs])), 1e-6) * 1e-3) / 1e9
    read_ceiling_best = max(sweep, key=sweep.get) if sweep else None
    read_ceiling = sweep[read_ceiling_best] if sweep else None
    flags_host = d_flags.cpu().numpy()

    # what each rank held and did, gathered for the driver (N = 1: one entry)
    ident.update({"frames_scanned": a.frames * a.steps, "kernel_ms": kern_ms, "wall_s": dt_local,
                  "motion_frames_in_batch": int(flags_host.sum()), "read_ceiling_GBps": read_ceiling})
    ranks = [ident]
    if multi:
        with open(rank_log_path(rank), "a") as f:
            f.write(json.dumps({"stage": "timed", **ident}) + "\n")
        ranks = [None] * world
        try:
            dist.all_gather_object(ranks, ident)
        except Exception as e:      # evidence only: the measured line must not be lost to it (each rank's log has its own record)
            ranks = [dict(ident, note="all_gather_object failed: " + repr(e))]

    if rank == 0:
        total_frames = a.frames * world * a.steps
        value = total_frames / dt
        # HBM bytes per launch: NOT measured by this run (PMC counters need rocprofv3 around the process) —
        # replayed from the committed summary of the builder's own --pmc passes over this same command
        traffic, traffic_source = replayed_traffic(a.workload, a.params, a.frames)
        traffic_detail = None
        measured = getattr(a, "pmc_result", None)
        if measured and measured[0] is not None:
            traffic_replayed = traffic
            traffic, traffic_source, traffic_detail = measured
            traffic_detail = dict(traffic_detail, replayed_figure_for_comparison=traffic_replayed)
        elif measured:
            traffic_source = (traffic_source or "no committed figure") + f" [not measured in this run: {measured[1]}]"
        cpu = None
        others = None
        host = None
        # the batch is the generated tile repeated: so must be its flags (checks every frame of the
        # 5 GB batch, not only the first tile, against the oracle-verified tile flags below)
        tile_flags = flags_host[: a.distinct]
        assert np.array_equal(flags_host, np.tile(tile_flags, reps)[: a.frames]), "flags are not tile-periodic"
        if world == 1 and a.cpu_seconds > 0:
            cpu = cpu_baseline(params, mv, off, tile_flags, a.cpu_seconds, a.workload)
        if world == 1 and not a.no_host and a.workload == "1080p_dense8x8":
            try:
                host = host_fed_leg(spec, mv, off, a.host_runs, a.host_seconds)
            except Exception as e:          # informational leg
                host = {"error": repr(e)}
        if host and cpu and "error" not in host:
            # the comparison a reader needs beside the host-fed figures: the CPU restatement scanning the same kind of
            # frames IN PLACE on the same CPU quota (cpu_baseline.value, `cores` threads) against this path, which
            # first moves them over PCIe — the GPU path wins on device-resident arrays, not on host-resident ones
            ref_cpu = cpu["value"]
            vs = {"cpu_baseline_frames_per_s": ref_cpu, "cpu_cores": cpu["cores"], "host_cpu_quota": cpu.get("host_cpu_quota"),
                  "hot_stream_compact8_zero_copy": host.get("compact8_zero_copy_frames_per_s", 0.0) / ref_cpu}
            for key, v in (host.get("config4_64_streams") or {}).items():
                if isinstance(v, dict) and v.get("frames_per_s_wall"):
                    vs[f"config4_{key}_wall"] = v["frames_per_s_wall"] / ref_cpu
                    if v.get("frames_per_s_steady"):
                        vs[f"config4_{key}_steady"] = v["frames_per_s_steady"] / ref_cpu
            vs["resident_value"] = value / ref_cpu
            host["vs_cpu_baseline_same_quota"] = vs
        if world == 1 and not a.no_others:
            del d_mv, d_off, w
            torch.cuda.empty_cache()
            try:
                others = other_workloads(dev, a.distinct, arena)
            except AssertionError:
                raise                       # a parity failure must fail the bench
            except Exception as e:          # e.g. out of memory on a smaller device: keep the headline
                others = [{"error": repr(e)}]
        roof = roofline_of(alg_bytes, kern_ms)
        # the planning kernels ahead of every scan (frames without side data answered, the others listed): their time,
        # and the rate of the whole call — what a caller of mtgpu_scan_frames_device gets per launch
        roof.update({"plan_ms": plan_ms, "call_ms": kern_ms + plan_ms,
                     "frac_of_call": alg_bytes / ((kern_ms + plan_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS})
        roof.update({"traffic": traffic, "traffic_source": traffic_source, "traffic_detail": traffic_detail,
                     "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                     "measured_read_ceiling": read_ceiling,
                     "measured_read_ceiling_is": f"best of a sweep of read-only kernels on the same buffer: {read_ceiling_best} "
                                                 "(load shape / bytes per workgroup [/ every n-th workgroup idle])",
                     "read_ceiling_sweep_GBps": {k_: round(v_, 1) for k_, v_ in sweep.items()},
                     "scan_rate_next_to_the_sweep_GBps": scan_in_sweep,
                     "frac_of_measured_ceiling": scan_in_sweep / read_ceiling if read_ceiling else None,
                     # rounds 1-4 quoted this ratio instead: the timed loop's rate over ONE read-only kernel, 16 contiguous
                     # bytes per lane, 1.25 MiB per workgroup (not the best of a sweep, not re-timed beside it)
                     "frac_of_16B_read_kernel_rounds_1_to_4_definition":
                         (roof["achieved"] / sweep["16B/1310720"]) if sweep.get("16B/1310720") else None})
        line = {
            "metric": "MV-scan frames/sec at 1080p grid" if a.workload.startswith("1080p") else "MV-scan frames/sec",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16/u32 (integer threshold + vote), f64 (segment merge)",
            "data": "synthetic",
            # (`workload` stays short: a record that keeps only its first ~100 characters must still show params=...)
            "config": {"workload": f"{a.workload} params={a.params} grid={params.grid_w}x{params.grid_h} "
                                   f"frames/GPU/step={a.frames}",
                       "params": a.params, "grid": [params.grid_w, params.grid_h], "records": "synthetic 40-byte AVMotionVector arrays",
                       "distinct_frames_tiled": a.distinct,
                       "frames_per_gpu": a.frames, "streams_per_gpu": S, "streams_total": S * world,
                       "records_per_step_per_gpu": n_records,
                       "bytes_per_step_per_gpu": alg_bytes, "parallelism": f"frame-sharded x{world}",
                       "step": "scan kernel" if a.no_merge else "scan + stream-merge kernels" +
                               (" + RCCL all_gather of segment lists" if multi else "")},
            "roofline": roof,
            "cpu_baseline": cpu,
            "other_workloads": others,
            "host_fed": host,
            "motion_frames_in_batch": int(flags_host.sum()),
            "ranks": ranks,
            "distinct_devices": len({(r or {}).get("pci_bus_id") or (r or {}).get("uuid") or i
                                     for i, r in enumerate(ranks)}) if len(ranks) == world else None,
        }
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if multi:
        stage("line_printed" if rank == 0 else "waiting_for_rank_0")
        dist.barrier()
        dist.destroy_process_group()
        stage("done")
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
        if hang["file"] is not None:
            hang["file"].close()
            hang["file"] = None
        try:                                   # the hang log of a run that did not hang is empty: drop it
            hang_path = rank_log_path(rank) + ".hang.log"
            if os.path.exists(hang_path) and os.path.getsize(hang_path) == 0:
                os.remove(hang_path)
        except OSError:
            pass
    scanner.close()
    del out


def cpu_quota():
    """CPUs' worth of time the cgroup grants this process (cpu.max / cfs quota), or None if unlimited / unknown:
    a box may show 256 hardware threads to a job that is allowed 16 of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def host_cpu_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count(), len(os.sched_getaffinity(0))


def cpu_baseline(params, mv, off, gpu_flags, budget_s, workload):
    """The C oracle (kind "port": our restatement of the reference's check_frame) timed on this host's
    cores on a bounded sample, as SURVEY.md 8(d) defines the CPU baseline: 1 thread, 16 threads (a 1-GPU
    box's CPU share) and ALL usable cores — frames split statically over pthreads, one private grid per
    thread (the reference's one-scanner-per-worker model, src/pipeline.cpp:186-197).  The sample is the
    `distinct` generated frames tiled so that every thread streams several MB per pass; each leg repeats
    the pass until its share of ~budget_s seconds of wall time is used.  `value` = the 16-thread leg."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob     # checker / baseline only
    model, cores_total, cores_usable = host_cpu_info()
    share = min(cores_usable, 16)
    n0 = len(off) - 1
    flags = ob.scan_frames(params, mv, off, None, nthreads=share)       # warm-up + parity check
    assert np.array_equal(flags, gpu_flags), "GPU flags differ from the oracle on the bench tile"
    # the sample: the distinct frames tiled to >= 8 frames (10 MB, beyond its L2) per thread of the widest leg.
    # Every thread scans a NUMA-local copy of its share (a worker's own decoder output in the reference),
    # `reps` times between two barriers: oracle/mt_oracle.c mto_bench_scan times exactly the scanning.
    want = max(n0, 8 * cores_usable)
    tile = max(1, (want + n0 - 1) // n0)
    counts = np.tile(np.diff(off.astype(np.int64)), tile)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    mv = np.tile(mv, tile)
    n = len(off) - 1

    def leg(threads, seconds):
        _, t1 = ob.bench_scan(params, mv, off, None, nthreads=threads, reps=1)       # calibration pass
        reps = int(max(1, min(100000, seconds / max(t1, 1e-6))))
        fl, dt = ob.bench_scan(params, mv, off, None, nthreads=threads, reps=reps)
        assert np.array_equal(fl, np.tile(gpu_flags, tile)[:n]), "oracle flags changed between passes"
        return n * reps / dt, reps, dt

    legs = [(1, 0.15 * budget_s), (share, 0.45 * budget_s)]
    if cores_usable > share:
        legs.append((cores_usable, 0.40 * budget_s))
    res = {t: leg(t, sec) for t, sec in legs}
    v16, reps16, dt16 = res[share]
    out = {"value": v16, "unit": "frames/s", "cores": share, "kind": "port",
           "sample": f"{n} {workload} frames ({n0} distinct, {mv.nbytes / 1e6:.0f} MB) x {reps16} passes "
                     f"({dt16:.1f} s between barriers), oracle/mt_oracle.c mto_bench_scan, {share} pthreads, "
                     f"each on a thread-local copy of its share",
           "value_1core": res[1][0], "sample_1core": f"{res[1][1]} passes, {res[1][2]:.1f} s",
           "host_cpu": model, "host_cores_total": cores_total, "host_cores_usable": cores_usable,
           "host_cpu_quota": cpu_quota(),     # CPUs of run time the cgroup allows (None: no limit): an all-thread leg
                                              # beyond it is time-sliced, not parallel
           "GBps": {"1": res[1][0] * mv.nbytes / n / 1e9, str(share): v16 * mv.nbytes / n / 1e9}}
    if cores_usable in res and cores_usable != share:
        va, ra, da = res[cores_usable]
        out.update({"value_all_cores": va, "cores_all": cores_usable,
                    "sample_all_cores": f"{ra} passes, {da:.1f} s, {cores_usable} pthreads (every usable hardware thread)"})
        out["GBps"][str(cores_usable)] = va * mv.nbytes / n / 1e9
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if needs_launch(a, os.environ):
        sys.exit(launch_ranks(a, argv))
    if a.pmc_child:
        return pmc_child(a)
    if a.gpus == 1 and "WORLD_SIZE" not in os.environ and not a.no_pmc and not a.force_dist:
        # N = 1, and this process has not touched the GPU yet (torch is not even imported)
        a.pmc_result = measure_traffic(a, argv)
    run_rank(a)


if __name__ == "__main__":
    main()
