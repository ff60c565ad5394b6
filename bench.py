#!/usr/bin/env python3
"""MV-scan bench (BASELINE.json metric: "MV-scan frames/sec at 1080p grid").

A step = one pass of the hot path over one device-resident batch of synthetic
1080p AVMotionVector arrays (config "Synthetic 1080p MV arrays, 120x68 16px grid,
30 fps stream", dense8x8 = 32 640 records = 1 305 600 B per P-frame):
    scan kernel (flags per frame)  ->  stream merge kernel (segments per stream)
    [N > 1: one RCCL all_gather of the per-GPU segment lists]
Inputs are resident in HBM before the timed region.  Frames shard across ranks
(weak scaling: every rank owns `--frames` frames of its own streams).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — the scan kernel: algorithmic bytes / live HIP-event duration vs 8 TB/s
  cpu_baseline — the C oracle timed on this host's cores on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--streams", type=int, default=8, help="streams per GPU (frames split evenly)")
    ap.add_argument("--distinct", type=int, default=60, help="distinct generated frames per GPU (tiled)")
    ap.add_argument("--workload", default="1080p_dense8x8",
                    choices=["1080p_dense8x8", "1080p_dense16", "4k_dense8x8", "4k_fine"])
    ap.add_argument("--params", default="code_defaults", choices=["code_defaults", "shipped_env"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--no-merge", action="store_true", help="time the scan kernel alone")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device rehearses the N>1 control flow on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def make_spec(workload, seed):
    from mvtrim_amd import synth
    if workload == "1080p_dense8x8":
        return synth.spec_1080p(seed=seed, sub=2), (1920, 1080, {})
    if workload == "1080p_dense16":
        return synth.spec_1080p(seed=seed, sub=1), (1920, 1080, {})
    if workload == "4k_dense8x8":
        return synth.spec_4k(seed=seed, sub=2), (3840, 2160, {})
    return synth.spec_4k_fine(seed=seed), (3840, 2160, dict(block_size=4, block_shift=2))


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    import mvtrim_amd as m
    from mvtrim_amd import dist as mdist
    from mvtrim_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if a.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    # ---------------- synthetic input: `distinct` generated frames, tiled to `frames`
    spec, (W, H, gridkw) = make_spec(a.workload, seed=1000 + rank)
    spec.events = synth.scripted_events(spec, a.distinct)
    mv, off, pts, sd = synth.gen_stream(spec, a.distinct)
    kw = dict(m.config.CODE_DEFAULTS if a.params == "code_defaults" else m.config.SHIPPED_ENV)
    kw.update(gridkw)
    if spec.sub == 1:
        kw["vectors_needed"] = 1      # one record per cell can never collect 2 votes in a cell
    params = m.ScanParams.from_config(W, H, **kw)
    scanner = m.MotionScanner(params, device=local)

    reps = (a.frames + a.distinct - 1) // a.distinct
    counts = np.tile(np.diff(off.astype(np.int64)), reps)[: a.frames]
    off_big = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    d_tile = torch.from_numpy(mv.view(np.uint8).copy()).to(dev)
    d_mv = d_tile.repeat(reps)[: int(off_big[-1]) * 40].contiguous()
    del d_tile
    d_off = torch.from_numpy(off_big).to(dev)
    d_flags = torch.empty(a.frames, dtype=torch.uint8, device=dev)
    n_records = int(off_big[-1])
    alg_bytes = 40 * n_records + 9 * a.frames          # 40*N_mv + 8 (offset) + 1 (flag) per frame

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

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    # two sets of merge outputs / gather buffers: the gather of step i overlaps the scan of step i+1
    packed_w = SEG_CAP * 16 + m.MERGE_RESULT_DTYPE.itemsize
    outs = [(torch.zeros((S, SEG_CAP, 2), dtype=torch.float64, device=dev),
             torch.zeros((S, m.MERGE_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev),
             torch.empty(2 * a.frames, dtype=torch.float64, device=dev)) for _ in range(2)]
    gath = [torch.empty((world, S, packed_w), dtype=torch.uint8, device=dev if a.backend == "nccl" else "cpu")
            for _ in range(2)] if world > 1 else None
    pending = [None, None]
    counter = [0]

    def step(i=None):
        k = counter[0] & 1
        counter[0] += 1
        if i is not None:
            ev0[i].record()
        scanner.check_frames_device(d_mv, d_off, None, d_flags)
        if i is not None:
            ev1[i].record()
        if a.no_merge:
            return None
        if pending[k] is not None:          # buffer set k is still being gathered (two steps ago)
            pending[k].wait()
            pending[k] = None
        seg, res = scanner.merge_streams_device(d_flags, d_pts, d_soff, d_mp, True, SEG_CAP, out=outs[k])
        if world > 1:
            # the only exchange step of the path: per-GPU segment lists to every rank (RCCL over xGMI),
            # asynchronous so that it overlaps the next step's scan
            packed = mdist.pack_segment_lists(seg, res)
            if a.backend == "nccl":
                pending[k] = dist.all_gather_into_tensor(gath[k], packed, async_op=True)
            else:       # rehearsal only: gloo moves the lists through host memory
                mdist.gather_segment_lists(seg.cpu(), res.cpu(), out=gath[k])
        return seg, res

    def finish():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for _ in range(a.warmup):
        step()
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(a.steps):
        out = step(i)
    finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in zip(ev0, ev1)]))
    # calibration (outside the timed region): pure read of the same record buffer, same load flavour
    lib = m.load_library()
    st = torch.cuda.current_stream(dev).cuda_stream
    nbytes = (d_mv.numel() // 16) * 16
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        m._abi.check(lib.mtgpu_debug_read_ceiling(scanner._ctx, d_mv.data_ptr(), nbytes, st))
    c0.record()
    for _ in range(10):
        m._abi.check(lib.mtgpu_debug_read_ceiling(scanner._ctx, d_mv.data_ptr(), nbytes, st))
    c1.record()
    torch.cuda.synchronize()
    read_ceiling = nbytes / (c0.elapsed_time(c1) / 10 * 1e-3) / 1e9
    flags_host = d_flags.cpu().numpy()

    if rank == 0:
        total_frames = a.frames * world * a.steps
        value = total_frames / dt
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes/launch from a --pmc run, if committed
        if os.path.exists(tp):
            try:
                rec = json.load(open(tp)).get(f"{a.workload}:{a.frames}")
                traffic = rec["hbm_bytes_per_launch"] if rec else None
            except Exception:
                traffic = None
        cpu = None
        # the batch is the generated tile repeated: so must be its flags (checks every frame of the
        # 5 GB batch, not only the first tile, against the oracle-verified tile flags below)
        tile_flags = flags_host[: a.distinct]
        assert np.array_equal(flags_host, np.tile(tile_flags, reps)[: a.frames]), "flags are not tile-periodic"
        if world == 1 and a.cpu_seconds > 0:
            cpu = cpu_baseline(params, mv, off, tile_flags, a.cpu_seconds, a.workload)
        line = {
            "metric": "MV-scan frames/sec at 1080p grid" if a.workload.startswith("1080p") else "MV-scan frames/sec",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16/u32 (integer threshold + vote), f64 (segment merge)",
            "data": "synthetic",
            "config": {"workload": f"synthetic {a.workload} MV arrays, {params.grid_w}x{params.grid_h} grid, "
                                   f"{a.frames} frames/GPU/step in {S} streams ({a.distinct} distinct frames tiled), "
                                   f"params={a.params}",
                       "frames_per_gpu": a.frames, "streams_per_gpu": S, "streams_total": S * world,
                       "records_per_step_per_gpu": n_records,
                       "bytes_per_step_per_gpu": alg_bytes, "parallelism": f"frame-sharded x{world}",
                       "step": "scan kernel" if a.no_merge else "scan + stream-merge kernels" +
                               (" + RCCL all_gather of segment lists" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "scan_frames_kernel", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes,
                         "measured_read_ceiling": read_ceiling, "frac_of_measured_ceiling": achieved / read_ceiling},
            "cpu_baseline": cpu,
            "motion_frames_in_batch": int(flags_host.sum()),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    scanner.close()
    del out


def cpu_baseline(params, mv, off, gpu_flags, budget_s, workload):
    """The C oracle (kind "port": our restatement of the reference's check_frame) timed on
    this host's cores on a bounded sample: the `distinct` generated frames, scanned
    repeatedly until ~budget_s seconds of wall time; frames split over all usable cores
    (one private grid per thread, the reference's one-scanner-per-worker model)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob     # checker / baseline only
    cores = min(len(os.sched_getaffinity(0)), 16)     # a 1-GPU box's CPU share is 16 cores
    n0 = len(off) - 1
    flags = ob.scan_frames(params, mv, off, None, nthreads=cores)       # warm-up + parity check
    assert np.array_equal(flags, gpu_flags), "GPU flags differ from the oracle on the bench tile"
    # tile the sample so that every thread streams tens of MB per pass (beyond its L2)
    tile = max(1, (64 * cores + n0 - 1) // n0)
    counts = np.tile(np.diff(off.astype(np.int64)), tile)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    mv = np.tile(mv, tile)
    n = len(off) - 1
    t0 = time.perf_counter()
    ob.scan_frames(params, mv, off, None, nthreads=1)
    t1 = time.perf_counter() - t0
    reps, t_mt = 0, 0.0
    t0 = time.perf_counter()
    while True:
        ob.scan_frames(params, mv, off, None, nthreads=cores)
        reps += 1
        t_mt = time.perf_counter() - t0
        if t_mt > budget_s or reps >= 10000:
            break
    return {"value": n * reps / t_mt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} {workload} frames ({n0} distinct, {mv.nbytes / 1e6:.0f} MB) x {reps} passes "
                      f"({t_mt:.1f} s wall), oracle/mt_oracle.c scan, {cores} pthreads",
            "value_1core": n / t1}


if __name__ == "__main__":
    main()
