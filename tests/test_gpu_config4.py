"""BASELINE config 4 at its stated width, through the product-shaped path (`-m gpu`):
64 distinct-seed 1080p dense8x8 streams as .mtmv files -> the C++ host layer's BatchProcessor re-plumb
(`process_batch` in csrc/host/mtgpu_host.hpp = /root/reference src/batch_processor.cpp:81-157, 307-350 with
GPUs in place of CPU sets) -> per-stream FFmpegJob segment lists, at 64 streams x 1 worker and at
16 streams x 4 workers.  Every job is compared bit for bit with a Python transcription of the reference's
worker loop driven by the oracle (chunks -> backward seek -> frame filter -> check_frame -> pool -> merge);
still streams must produce no job; a damaged file must fail alone.  The batch summary line reports what
64 x T workers hold on one device (one shared context, 64 x T pipes, HIP streams, pinned bytes).
"""
import json
import os
import subprocess

import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import synth

import oracle_binding as ob
from test_gpu_parity import _check_job, _reference_worker_loop

pytestmark = pytest.mark.gpu

N_STREAMS = 64
ENV_CFG = dict(CHUNK_DURATION_SEC="0.75", TARGET_FPS="15", MAX_GAP_SEC="0.4", PADDING_SEC="0.1",
               MIN_SAVINGS_PCT="5")
STILL = (7, 41)                      # streams without any event (and without salt noise): "No motion found"


def stream_events(spec, n, rng):
    """1-3 short events per stream so that segments, gaps on both sides of MAX_GAP_SEC and the
    frame-skip filter all matter inside a 2-3 s stream."""
    ev, f = [], int(rng.randint(2, 12))
    for _ in range(int(rng.randint(1, 4))):
        if f >= n - 3:
            break
        length = int(rng.randint(3, 18))
        cw, ch = int(rng.randint(2, 7)), int(rng.randint(2, 6))
        cx = int(rng.randint(1, spec.cells_x - cw - 1))
        cy = int(rng.randint(spec.cells_y // 10 + 1, spec.cells_y - ch - spec.cells_y // 10 - 1))
        ev.append(synth.Event(f, min(n, f + length), cx, cy, cw, ch, int(rng.choice([-1, 1]) * rng.randint(4, 13)),
                              int(rng.randint(-6, 7))))
        f += length + int(rng.choice([4, 9, 16, 30]))
    return ev


@pytest.fixture(scope="module")
def streams64(tmp_path_factory):
    """64 distinct-seed 1080p dense8x8 .mtmv streams (66-90 frames each, ~6 GB) + the oracle-driven answers."""
    d = tmp_path_factory.mktemp("config4")
    p = ob.params_from_config(1920, 1080)                  # code defaults: T=16, VECTORS_NEEDED=2, CLUSTERS_NEEDED=2
    cases, paths = {}, []
    rng = np.random.RandomState(64)
    for k in range(N_STREAMS):
        n = int(rng.randint(66, 91))
        still = k in STILL
        spec = synth.spec_1080p(seed=4000 + k, sub=2, salt_p=0.0 if still else 1e-3)
        spec.events = [] if still else stream_events(spec, n, rng)
        frames = [synth.gen_frame(spec, i) for i in range(n)]
        ticks = [spec.pts_ticks(i) for i in range(n)]
        path = str(d / f"cam{k:02d}.mtmv")
        m.mvfile.write_mtmv(path, 1920, 1080, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
        mp = m.MergeParams(duration=n / spec.fps, max_gap_sec=0.4, padding_sec=0.1, min_savings_pct=5.0)
        pooled, (want_seg, want_res) = _reference_worker_loop(spec, frames, ticks, n / spec.fps, p, 0.75, 15.0, mp)
        scanned = sum(len(ob.filter_frames(ticks[first:], 1.0 / spec.tb_den, c0, c1, 2)[0])
                      for (c0, c1, _), first in _chunk_starts(spec, frames, ticks, n / spec.fps, 0.75))
        cases[path] = (pooled, want_seg, want_res, scanned)
        paths.append(path)
        del frames
    yield str(d), paths, cases
    import shutil
    shutil.rmtree(str(d), ignore_errors=True)              # 6 GB of streams: do not leave them to pytest's retention


def _chunk_starts(spec, frames, ticks, duration, chunk_sec):
    tb = 1.0 / spec.tb_den
    keys = [i for i, f in enumerate(frames) if f is None]
    for c in m.make_chunks(duration, chunk_sec):
        target = int(c[0] / tb)
        yield c, max([k for k in keys if ticks[k] <= target] or [0])


def _run(paths, streams, threads, outdir, extra_env=None):
    exe = os.path.join(os.path.dirname(m.LIB_PATH), "mtgpu_scan_file")
    assert os.path.exists(exe), "build it with make -C motion-estimated-video-trimmer_amd/csrc"
    env = dict(os.environ, **ENV_CFG)
    for k in ("MV_THRESHOLD_SQ", "VECTORS_NEEDED", "CLUSTERS_NEEDED", "BLOCK_SIZE", "BLOCK_SHIFT", "VERTICAL_MASK",
              "MTGPU_STAGING", "MTGPU_BATCH_MB"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([exe] + paths + ["--streams", str(streams), "--threads", str(threads), "--outdir", outdir,
                                        "--summary"], capture_output=True, text=True, env=env, timeout=600)
    lines = [json.loads(ln) for ln in r.stdout.strip().splitlines()]
    jobs = [j for j in lines if "batch_summary" not in j]
    summary = [j["batch_summary"] for j in lines if "batch_summary" in j]
    assert len(summary) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return r, jobs, summary[0]


def _check_all(jobs, cases, paths):
    want_jobs = sorted(pth for pth in paths if cases[pth][2]["do_cut"] >= 0)
    assert sorted(j["input"] for j in jobs) == want_jobs
    for j in jobs:
        pooled, want_seg, want_res, scanned = cases[j["input"]]
        _check_job(j, pooled, want_seg, want_res)
        assert j["frames_scanned"] == scanned


@pytest.mark.parametrize("streams,threads", [(64, 1), (16, 4)])
def test_config4_64_streams_through_cpp_host_layer(streams64, streams, threads):
    d, paths, cases = streams64
    for k in STILL:
        assert cases[paths[k]][2]["do_cut"] == -1              # the still streams: no motion, no job
    assert sum(1 for pth in paths if len(cases[pth][1]) >= 2) >= 10     # many streams really have several segments
    r, jobs, s = _run(paths, streams, threads, d)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_all(jobs, cases, paths)
    assert (s["streams"], s["threads_per_stream"], s["videos"], s["failed"]) == (streams, threads, N_STREAMS, 0)
    assert s["jobs"] == len(jobs) == N_STREAMS - len(STILL)
    assert s["frames_scanned"] == sum(c[3] for c in cases.values())
    # ONE context for the whole process (same device, same parameter block: SharedContext) and one pinned pipe
    # (3 staging batches, each with its stream and event, one pinned slab) per worker thread
    h = s["held"]
    assert h["contexts"] == 1 and h["mem_pools"] == 1 and h["pipes"] == streams * threads
    # the pipes' batches run on the context's pool of 8 streams (creating a stream costs ~3.5 ms, serialised by the
    # runtime: 192 of them were the bulk of pipe set-up), events stay per batch
    assert h["hip_streams"] == 8 + 1 and h["hip_events"] == streams * threads * 3
    # staging is page-locked per batch on first use: every worker pinned at least its first batch, none more than 3
    assert streams * threads * (16 << 20) <= h["pinned_bytes"] <= streams * threads * 3 * (16 << 20) * 1.05
    assert streams * threads <= h["pinned_batches"] <= 3 * streams * threads
    print(f"\nconfig 4, {streams} streams x {threads} workers on one device: {s['frames_scanned']} frames in "
          f"{s['wall_us'] / 1e3:.0f} ms wall; held {h['contexts']} contexts, {h['hip_streams']} HIP streams, "
          f"{h['pinned_bytes'] / 2**20:.0f} MiB pinned, {h['device_bytes'] / 2**20:.1f} MiB device")


def test_config4_damaged_file_fails_alone(streams64, tmp_path):
    """A truncated file and a file with a damaged frame table among the 64: each fails by itself (exit code 1,
    one error line each), every other stream's job is still bit-exact."""
    d, paths, cases = streams64
    raw = open(paths[3], "rb").read(4 << 20)
    trunc = str(tmp_path / "truncated.mtmv")
    open(trunc, "wb").write(raw)                               # header promises more records than the file holds
    bad = bytearray(raw[:1 << 20])
    n_frames = int(np.frombuffer(raw, dtype="<u8", count=1, offset=40)[0])
    bad[32:40] = np.float64(2.0).tobytes()                     # duration
    bad[48:56] = np.uint64(100).tobytes()                      # n_records = 100 ...
    tab = str(tmp_path / "badtable.mtmv")
    open(tab, "wb").write(bytes(bad[:56 + 24 * n_frames + 4000]))   # ... but frame 1 still claims 32 640 of them
    mixed = paths[:20] + [trunc] + paths[20:40] + [tab] + paths[40:]
    r, jobs, s = _run(mixed, 64, 1, d)
    assert r.returncode == 1
    errs = [ln for ln in r.stderr.splitlines() if ln.startswith("error:")]
    assert len(errs) == 2 and any("truncated" in e for e in errs) and any("frame table" in e for e in errs)
    assert s["failed"] == 2 and s["videos"] == N_STREAMS + 2
    _check_all(jobs, cases, paths)


def test_config4_bad_environment_is_an_error_not_a_crash(streams64):
    """a value that does not parse used to escape a worker thread (std::terminate)."""
    d, paths, _ = streams64
    r, jobs, s = _run(paths[:4], 4, 1, d, {"MTGPU_BATCH_MB": "abc"})
    assert r.returncode == 1 and jobs == [] and s["failed"] == 4
    assert r.stderr.count("configuration:") == 4


def test_failed_video_does_not_hand_its_pipe_to_the_next_one(streams64):
    """a worker's backend (context + pinned pipe) outlives a video (process_batch keeps it for the next
    file).  When a collect fails — here injected ONCE in the process, with the batch's drain failing too, so the
    batch is retired — the video fails, and the pipe, whose state is unknown, must not serve the next video: the
    backend is marked dirty and rebuilt.  One stream x one worker over three files: exactly one fails, the other
    two are bit-exact, one pipe was rebuilt."""
    d, paths, cases = streams64
    three = [pth for pth in paths if cases[pth][2]["do_cut"] >= 0][:3]
    r, jobs, s = _run(three, 1, 1, d, {"MTGPU_INJECT_COLLECT_FAIL": "-2", "MTGPU_INJECT_ONCE": "1"})
    assert r.returncode == 1 and s["failed"] == 1 and s["videos"] == 3
    errs = [ln for ln in r.stderr.splitlines() if ln.startswith("error:")]
    assert len(errs) == 1 and three[0] in errs[0] and "retired" in errs[0]
    assert sorted(j["input"] for j in jobs) == sorted(three[1:])
    for j in jobs:
        pooled, want_seg, want_res, scanned = cases[j["input"]]
        _check_job(j, pooled, want_seg, want_res)
        assert j["frames_scanned"] == scanned
    assert s["held"]["pipe_rebuilds"] == 1 and s["held"]["pipes"] == 1


def test_multi_device_paths_of_the_host_layer_on_aliased_devices(streams64):
    """VERDICT r3 "missing 4": the in-process multi-device path of the C++ host layer (worker_device round-robin,
    one SharedContext + scratch pool per (device, parameters), pipes created on their worker's device; reference
    stream -> CPU-set assignment src/batch_processor.cpp:102-110) had only a table test.  MTGPU_ALIAS_DEVICES=4
    makes the library present four logical devices (all backed by the one physical GPU of this box), so
    process_batch at 8 streams x 2 workers takes exactly the code path an N-GPU node takes: four contexts, every
    job still bit-exact.  (Hardware with several GPUs is the driver's; this pins the logic.)"""
    d, paths, cases = streams64
    some = paths[:24]
    r, jobs, s = _run(some, 8, 2, d, {"MTGPU_ALIAS_DEVICES": "4"})
    assert r.returncode == 0, r.stderr[-2000:]
    _check_all(jobs, cases, some)
    h = s["held"]
    assert h["contexts"] == 4 and h["mem_pools"] == 4 and h["pipes"] == 16        # one context + pool per logical device
    r1, jobs1, s1 = _run(some, 8, 2, d)
    assert s1["held"]["contexts"] == 1
    assert sorted(json.dumps(j["segments"]) + j["input"] for j in jobs) == sorted(json.dumps(j["segments"]) + j["input"] for j in jobs1)


def test_host_layer_repeated_runs_stay_bit_exact(streams64):
    """The same 24 videos through process_batch again and again (MTGPU_STRESS_RUNS, default 6; a soak sets it higher):
    16 workers on four aliased contexts, then 8 x 1 on one — tiny batches (10-20 frames per submit) from many threads on
    the context's shared streams, each launch = planning kernel + scan kernel.  Round 6 saw ONE job of this shape with
    one motion frame missing, once (gpurun_out/r06/dense_fast_path_1.log), while launches took their work-list memory from
    the context's stream-ordered pool; a batch now owns that memory (mtgpu_pipe_stats.list_bytes).  Every job of every
    run is compared with the oracle-driven answer."""
    d, paths, cases = streams64
    some = paths[:24]
    for it in range(int(os.environ.get("MTGPU_STRESS_RUNS", "6"))):
        for streams, threads, env in ((8, 2, {"MTGPU_ALIAS_DEVICES": "4"}), (8, 1, None)):
            r, jobs, s = _run(some, streams, threads, d, env)
            assert r.returncode == 0, (it, r.stderr[-2000:])
            _check_all(jobs, cases, some)


def test_batch_sized_from_budget_devices_and_videos_when_no_counts_are_given(streams64):
    """`mtgpu_scan_file --streams 0 --threads 0` sizes the batch with the host layer's own rule (default_batch_sizing:
    CPU budget, devices, videos — hand cases in tests/test_reference_host.py), honouring PARALLEL_STREAMS and
    THREADS_PER_STREAM as `motion_trim in_dir out_dir` does (config.hpp:138-141, 165-168); the jobs stay bit-exact and
    an unparsable value is a configuration error, not a crash."""
    d, paths, cases = streams64
    some = paths[:8]
    r, jobs, s = _run(some, 0, 0, d, {"PARALLEL_STREAMS": "4", "THREADS_PER_STREAM": "2"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert (s["streams"], s["threads_per_stream"]) == (4, 2)
    _check_all(jobs, cases, some)
    exe = os.path.join(os.path.dirname(m.LIB_PATH), "mtgpu_scan_file")
    r = subprocess.run([exe] + some + ["--streams", "0", "--threads", "0", "--outdir", d], capture_output=True, text=True,
                       env=dict(os.environ, PARALLEL_STREAMS="abc"), timeout=120)
    assert r.returncode == 1 and "configuration" in r.stderr and r.stdout == ""
