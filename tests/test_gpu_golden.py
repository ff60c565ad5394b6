"""GPU tier, part 2 (`-m gpu`): the committed golden fixtures and BASELINE.json's multi-stream /
multi-GPU configurations DIRECTLY through the HIP path (C ABI of libmtgpu.so).

  * tests/golden/check_frame_hand_cases.json   -> mtgpu_scan_frames, every counter form, both
                                                  record layouts (40-byte AoS, 8-byte compact)
  * tests/golden/merge_hand_cases.json         -> mtgpu_merge_segments / mtgpu_merge_streams_device
  * tests/golden/survey_segments.json          -> mtgpu_merge_segments, 17-digit prints
  * tests/golden/frame_filter_hand_cases.json  -> Python host mirror + C++ GpuMotionScanner
  * config 4 (64 concurrent 1080p streams)     -> scan -> merge_streams_device -> pack ->
                                                  mtgpu_gather_segments, per-stream vs the oracle
  * time-range split of one stream at world 2 / 8 (ranks simulated one after the other)
  * extract_mvs JSON front end (mvjson)        -> HIP scan
"""
import json
import os
import subprocess

import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import dist as mdist
from mvtrim_amd import synth

import oracle_binding as ob
from conftest import experiments_build
from golden_cases import (build_mvs, id_of, load_filter_cases, load_hand_cases, load_merge_cases,
                          load_survey_segments, merge_case_ts)

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def scan_compact(scanner, mv, off, has_sd):
    """Frames through the compact-record entry point: library packer -> device -> REC 8 kernel."""
    import torch
    rec = m.pack_records(mv)
    d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda() if len(rec) else \
        torch.zeros(8, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint64).astype(np.int64)).cuda()
    d_sd = None if has_sd is None else torch.from_numpy(np.ascontiguousarray(has_sd, dtype=np.uint8)).cuda()
    got = scanner.check_frames_device_compact(d_rec[: len(rec) * 8], d_off, d_sd)
    torch.cuda.synchronize()
    return got.cpu().numpy()


# ------------------------------------------------------------------ check_frame hand cases

@pytest.mark.parametrize("force_fb", [None, 2, 8, 108])
@pytest.mark.parametrize("name,kw,case", load_hand_cases()[1], ids=id_of)
def test_check_frame_hand_cases_on_gpu(gpu_scanner_factory, name, kw, case, force_fb):
    """Every hand-derived known answer (reference src/motion_scanner.cpp:217-295) through
    mtgpu_scan_frames: 32-bit add counters (None), 2-/8-bit thermometer fields, 8-bit CAS fields."""
    p = m.ScanParams.from_config(**kw)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    if force_fb is not None:
        assert s.plan["counter_bits"] == force_fb % 100
    mv = build_mvs(case)
    sd = int(case.get("has_sd", 1))
    b = m.FrameBatch.from_frames([mv if sd else None])
    if not sd:                                         # "no side data whatever the records": keep the records
        b = m.FrameBatch(mv, np.array([0, len(mv)], dtype=np.uint64), None, np.zeros(1, dtype=np.uint8))
    got = s.check_frames(b)
    assert got.tolist() == [case["expect"]], name
    assert scan_compact(s, b.mv, b.frame_off, b.has_sd).tolist() == [case["expect"]], name
    if len(mv) > 1:                                    # record order must not matter
        b2 = m.FrameBatch(mv[::-1].copy(), b.frame_off, None, b.has_sd)
        assert s.check_frames(b2).tolist() == [case["expect"]]


def test_check_frame_hand_cases_one_batch(gpu_scanner_factory):
    """All cases that share the base parameters in ONE batch (frames of very different sizes
    side by side, incl. empty and side-data-less ones), sliced and unsliced."""
    g, cases = load_hand_cases()
    base = [(n, c) for n, kw, c in cases if kw == g["base"]]
    assert len(base) >= 15
    p = m.ScanParams.from_config(**g["base"])
    frames = [build_mvs(c) if int(c.get("has_sd", 1)) else None for _, c in base]
    want = [c["expect"] if int(c.get("has_sd", 1)) else 0 for _, c in base]
    b = m.FrameBatch.from_frames(frames)
    for slices in (1, 4):
        s = gpu_scanner_factory(p)
        s.set_slices(slices)
        assert s.check_frames(b).tolist() == want
        assert scan_compact(s, b.mv, b.frame_off, b.has_sd).tolist() == want


# ------------------------------------------------------------------ merge hand cases

@pytest.mark.parametrize("name,kw,case", load_merge_cases(), ids=id_of)
def test_merge_hand_cases_on_gpu(gpu_scanner_factory, name, kw, case):
    """Hand-derived merge / clamp / savings / decision answers (src/pipeline.cpp:302-358, 387-388)
    through mtgpu_merge_segments and mtgpu_merge_streams_device: exact doubles."""
    import torch
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080))
    ts = merge_case_ts(case)
    mp = m.MergeParams(**kw)
    seg, res = s.merge_segments(ts, mp, False)
    assert [list(x) for x in seg.tolist()] == case["segments"]
    assert res["time_removed"] == case["time_removed"] and res["do_cut"] == case["do_cut"]
    if "n_unique" in case:
        assert res["n_timestamps"] == case["n_unique"]
    job, jres = s.merge_segments(ts, mp, True)
    assert [list(x) for x in job.tolist()] == case["job"] and jres["n_segments"] == len(case["job"])
    # the device-resident stream entry point: the timestamps as one stream of all-flagged frames
    n = len(ts)
    d_pts = torch.tensor(ts if n else [0.0], dtype=torch.float64, device="cuda")
    d_fl = torch.ones(max(n, 1), dtype=torch.uint8, device="cuda")
    if n == 0:
        d_fl.zero_()
    soff = torch.tensor([0, max(n, 1)], dtype=torch.int64, device="cuda")
    d_mp = torch.from_numpy(mp.to_record().view(np.uint8).copy()).cuda()
    dseg, dres = s.merge_streams_device(d_fl, d_pts, soff, d_mp, job_semantics=True, seg_cap=128)
    torch.cuda.synchronize()
    rec = m.results_from_bytes(dres.cpu().numpy())[0]
    k = int(rec["n_segments"])
    assert k == len(case["job"]) and int(rec["do_cut"]) == case["do_cut"]
    assert dseg[0, :k].cpu().numpy().tolist() == case["job"]
    assert float(rec["time_removed"]) == case["time_removed"]


def test_merge_reproduces_survey_recorded_segments_on_gpu(gpu_scanner_factory):
    """tests/golden/survey_segments.json: the two FFmpegJob segments SURVEY.md section 8c recorded
    from a run of the reference's object code; the device merge prints the same 17 digits."""
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080))
    g, ts, mp = load_survey_segments()
    for order in (ts, ts[::-1], list(np.random.RandomState(1).permutation(ts))):
        seg, res = s.merge_segments(order, mp, True)
        assert res["do_cut"] == 1
        assert [["%.17g" % a, "%.17g" % b] for a, b in seg.tolist()] == g["segments_printed_17g"]


def test_reference_documented_examples_on_gpu(gpu_scanner_factory):
    """The behaviour the reference documents in config/motion_trim.env
    (tests/golden/reference_documented_examples.json) through the HIP path."""
    import documented_examples as de
    scanners = {}

    def check_frames(p, frames):
        key = (p.mv_threshold_sq, p.vectors_needed, p.clusters_needed, p.vertical_margin, p.block_shift)
        if key not in scanners:
            scanners[key] = gpu_scanner_factory(p)
        return scanners[key].check_frames(m.FrameBatch.from_frames(frames))
    s0 = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080))
    de.run(lambda w, h, **kw: m.ScanParams.from_config(w, h, **kw), check_frames,
           lambda ts, mp, job: s0.merge_segments(ts, mp, job), m.frame_skip)


# ------------------------------------------------------------------ a6 / a7 hand cases

def _filter_golden_stream(tmp_path):
    """The `pipeline` case of frame_filter_hand_cases.json as an .mtmv file: 100 frames at
    pts i/32 (time_base 1/64), keyframes without side data at 0/20/50, every other frame with a
    strong 2x2-cell cluster."""
    g = load_filter_cases()
    pl = g["pipeline"]
    frames = []
    for i in range(pl["n_frames"]):
        if i in pl["keyframes"]:
            frames.append(None)
            continue
        mv = np.zeros(8, dtype=m.MV_DTYPE)
        mv["dst_x"] = [168, 168, 184, 184, 168, 168, 184, 184]
        mv["dst_y"] = [120, 120, 120, 120, 136, 136, 136, 136]
        mv["src_x"] = mv["dst_x"] - 9
        mv["src_y"] = mv["dst_y"]
        mv["w"], mv["h"], mv["motion_scale"], mv["source"] = 16, 16, 4, -1
        frames.append(mv)
    ticks = [g["ticks_per_frame"] * i for i in range(pl["n_frames"])]
    path = str(tmp_path / "filter.mtmv")
    m.mvfile.write_mtmv(path, 320, 240, 1, g["time_base_den"], pl["fps"], pl["duration"], ticks, frames)
    return g, pl, frames, ticks, path


def test_frame_filter_hand_cases_python_host(gpu_scanner_factory, tmp_path):
    """MotionScanner.scan_range (Python host mirror + HIP check_frame) on the hand-derived
    whole-video case: chunk by chunk, then pooled."""
    g, pl, frames, ticks, _ = _filter_golden_stream(tmp_path)
    s = gpu_scanner_factory(m.ScanParams.from_config(320, 240))
    tb = 1.0 / g["time_base_den"]
    pooled = []
    for k, (c0, c1, _) in enumerate(m.make_chunks(pl["duration"], pl["chunk_sec"])):
        target = int(c0 / tb)
        first = max(f for f in pl["keyframes"] if ticks[f] <= target) if c0 > 0 else 0
        got = s.scan_range(ticks[first:], frames[first:], tb, c0, c1, pl["fps"], target_fps=pl["target_fps"])
        assert got == [f / 32.0 for f in pl["per_chunk"][k] if f not in pl["keyframes"]]
        pooled += got
    assert pooled == [f / 32.0 for f in pl["timestamps_frames"]]


@pytest.mark.parametrize("staging", ["compact8", "aos40", "compact8_zc", "aos40_zc"])
def test_frame_filter_hand_cases_cpp_host(tmp_path, staging):
    """The same case through the C++ host layer (GpuMotionScanner::scan_range, chunk workers,
    MtmvSource's backward seek): pooled timestamps == the hand-derived list, any worker count."""
    g, pl, frames, ticks, path = _filter_golden_stream(tmp_path)
    exe = os.path.join(os.path.dirname(m.LIB_PATH), "mtgpu_scan_file")
    env = dict(os.environ, CHUNK_DURATION_SEC=str(pl["chunk_sec"]), TARGET_FPS=str(pl["target_fps"]),
               MAX_GAP_SEC="5.0", PADDING_SEC="0.5", MIN_SAVINGS_PCT="5", MTGPU_STAGING=staging)
    for k in ("VECTORS_NEEDED", "CLUSTERS_NEEDED", "MV_THRESHOLD_SQ", "BLOCK_SIZE", "BLOCK_SHIFT", "VERTICAL_MASK"):
        env.pop(k, None)
    for threads in (1, 2, 3):
        out = subprocess.run([exe, path, "--threads", str(threads), "--timestamps"], check=True,
                             capture_output=True, text=True, env=env).stdout
        r = json.loads(out)
        assert r["chunks"] == 3
        assert r["timestamps"] == [f / 32.0 for f in pl["timestamps_frames"]]
        # one segment: first 2/32 - 0.5 -> clamped to 0, last 94/32 + 0.5 -> clamped to duration 3.0
        assert r["segments"] == [[0.0, 3.0]] and r["do_cut"] == 0


# ------------------------------------------------------------------ config 4: 64 concurrent 1080p streams

def _gen_streams(n_streams, n_frames, seed0):
    """n_streams distinct-seed 1080p dense8x8 streams, each with its own scripted events."""
    from concurrent.futures import ThreadPoolExecutor

    def one(i):
        spec = synth.spec_1080p(seed=seed0 + i, sub=2, gop=16, salt_p=0.0 if i % 16 == 5 else 1e-3)
        rng = np.random.RandomState(seed0 + i)
        ev, f = [], int(rng.randint(1, 6))
        while f < n_frames:                                # short bursts, gaps on both sides of max_gap
            ln = int(rng.randint(1, 5))
            ev.append(synth.Event(f, min(n_frames, f + ln), int(rng.randint(2, 100)), int(rng.randint(8, 50)),
                                  int(rng.randint(2, 6)), int(rng.randint(2, 5)), int(rng.choice([-9, 7, 12])), 2))
            f += ln + int(rng.choice([2, 3, 9, 14]))
        if i % 16 == 5:
            ev = []                                        # a still stream: no motion at all
        spec.events = ev
        return spec, synth.gen_stream(spec, n_frames)

    with ThreadPoolExecutor(max_workers=8) as ex:
        return list(ex.map(one, range(n_streams)))


def test_config4_64_streams_scan_merge_gather(gpu_scanner_factory):
    """BASELINE.json config 4 on one GPU: 64 distinct-seed 1080p dense8x8 streams in one device
    batch -> mtgpu_scan_frames_device -> mtgpu_merge_streams_device -> pack_segment_lists ->
    1-rank mtgpu_gather_segments (RCCL) -> per-stream bit-compare with the oracle.  Then the same
    streams as 8 ranks x 8 streams (ranks run one after the other on this GPU, their packed blocks
    concatenated rank-major exactly as the all-gather delivers them)."""
    import ctypes as C
    import torch
    S, F, CAP = 64, 32, 32
    streams = _gen_streams(S, F, seed0=4000)
    p = ob.params_from_config(1920, 1080)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080))
    # (np.concatenate would repack the padded 40-byte record dtype into 32 bytes: fill a typed array)
    mv = np.zeros(sum(len(st[1][0]) for st in streams), dtype=m.MV_DTYPE)
    pos = 0
    for st in streams:
        mv[pos:pos + len(st[1][0])] = st[1][0]
        pos += len(st[1][0])
    assert mv.dtype.itemsize == 40
    counts = np.concatenate([np.diff(st[1][1].astype(np.int64)) for st in streams])
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    pts = np.concatenate([st[1][2] for st in streams])
    sd = np.concatenate([st[1][3] for st in streams])
    mps = [m.MergeParams(duration=F / 30.0, max_gap_sec=0.2, padding_sec=0.05, min_savings_pct=5.0) for _ in range(S)]
    want_flags = ob.scan_frames(p, mv, off.astype(np.uint64), sd, nthreads=8)
    assert 0.1 < want_flags.mean() < 0.9

    d_mv = torch.from_numpy(mv.view(np.uint8).reshape(-1)).cuda()
    d_off = torch.from_numpy(off).cuda()
    d_sd = torch.from_numpy(sd).cuda()
    d_pts = torch.from_numpy(pts).cuda()
    soff = torch.from_numpy(np.arange(S + 1, dtype=np.int64) * F).cuda()
    d_mp = torch.from_numpy(np.concatenate([x.to_record() for x in mps]).view(np.uint8).copy()).cuda()
    flags = s.check_frames_device(d_mv, d_off, d_sd)
    seg, res = s.merge_streams_device(flags, d_pts, soff, d_mp, job_semantics=True, seg_cap=CAP)
    torch.cuda.synchronize()
    got_flags = flags.cpu().numpy()
    bad = np.flatnonzero(got_flags != want_flags)
    assert bad.size == 0, (bad.size, bad[:16], got_flags[bad[:16]], want_flags[bad[:16]], int(got_flags.sum()), int(want_flags.sum()))
    packed = mdist.pack_segment_lists(seg, res)

    lib = m.load_library()
    uid = (C.c_char * 128)()
    m._abi.check(lib.mtgpu_comm_unique_id(uid))
    comm = C.c_void_p()
    m._abi.check(lib.mtgpu_comm_create(0, 1, uid, 0, C.byref(comm)))
    try:
        recv = torch.zeros((1,) + tuple(packed.shape), dtype=torch.uint8, device="cuda")
        m._abi.check(lib.mtgpu_gather_segments(comm, packed.data_ptr(), packed.numel(), recv.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        lib.mtgpu_comm_destroy(comm)
    lists = mdist.assemble_stream_lists(recv, CAP, [S])

    def check_lists(lists):
        assert len(lists) == S
        n_cut = n_still = 0
        for i, ent in enumerate(lists):
            a, b = i * F, (i + 1) * F
            want_seg, want_res = ob.pool_and_merge(pts[a:b][want_flags[a:b] != 0], mps[i], True)
            r = ent["result"]
            assert int(r["status"]) == 0 and int(r["n_timestamps"]) == want_res["n_timestamps"], i
            assert int(r["n_segments"]) == want_res["n_segments"] and int(r["do_cut"]) == want_res["do_cut"], i
            assert np.array_equal(bits(ent["segments"][:, 0]), bits(want_seg["start"])), i
            assert np.array_equal(bits(ent["segments"][:, 1]), bits(want_seg["end"])), i
            assert bits([r["time_removed"], r["saved_pct"]]).tolist() == \
                bits([want_res["time_removed"], want_res["saved_pct"]]).tolist(), i
            n_cut += int(r["do_cut"]) == 1
            n_still += int(r["do_cut"]) == -1
        assert n_cut >= 32 and n_still == 4
    check_lists(lists)

    # 8 ranks x 8 streams: every rank scans + merges only its own streams' frames
    blocks = []
    for rank in range(8):
        a, b = mdist.shard_range(S, 8, rank)
        fa, fb = a * F, b * F
        ra, rb = int(off[fa]), int(off[fb])
        r_off = torch.from_numpy(off[fa:fb + 1] - off[fa]).cuda()
        r_flags = s.check_frames_device(d_mv[ra * 40: rb * 40], r_off, d_sd[fa:fb])
        r_soff = torch.from_numpy(np.arange(b - a + 1, dtype=np.int64) * F).cuda()
        r_seg, r_res = s.merge_streams_device(r_flags, d_pts[fa:fb], r_soff, d_mp[a * 32: b * 32].contiguous(),
                                              job_semantics=True, seg_cap=CAP)
        blocks.append(mdist.pack_segment_lists(r_seg, r_res))
    gathered = torch.stack(blocks)                                # [world, S/world, cap*16+40]: rank-major
    torch.cuda.synchronize()
    check_lists(mdist.assemble_stream_lists(gathered, CAP, [8] * 8))


# ------------------------------------------------------------------ time-range split of one stream

@pytest.mark.parametrize("world", [2, 8])
def test_timerange_split_world_simulated(gpu_scanner_factory, world):
    """One stream cut into `world` contiguous frame ranges of ~equal record counts
    (dist.shard_by_records); every range is scanned by the HIP path as its rank would, the ranks'
    timestamps are pooled rank-major (what gather_timestamps returns) and merged ONCE: segments
    bit-identical to the whole-stream result and to the oracle (src/pipeline.cpp:302-356)."""
    import torch
    spec = synth.spec_1080p(seed=77, sub=2)
    n = 240
    spec.events = synth.scripted_events(spec, n)
    mv, off, pts, sd = synth.gen_stream(spec, n)
    p = ob.params_from_config(1920, 1080)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080))
    mp = m.MergeParams(duration=n / 30.0, max_gap_sec=1.0, padding_sec=0.25, min_savings_pct=5.0)
    want_flags = ob.scan_frames(p, mv, off, sd, nthreads=8)
    want_seg, want_res = ob.pool_and_merge(pts[want_flags != 0], mp, True)
    assert want_res["n_segments"] >= 2

    d_mv = torch.from_numpy(mv.view(np.uint8).reshape(-1)).cuda()
    off64 = off.astype(np.int64)
    pooled, got_flags = [], []
    for (a, b) in mdist.shard_by_records(off64, world):
        if b == a:
            continue
        ra, rb = int(off64[a]), int(off64[b])
        fl = s.check_frames_device(d_mv[ra * 40: rb * 40], torch.from_numpy(off64[a:b + 1] - off64[a]).cuda(),
                                   torch.from_numpy(sd[a:b]).cuda()).cpu().numpy()
        got_flags.append(fl)
        pooled.append(pts[a:b][fl != 0])
    assert np.array_equal(np.concatenate(got_flags), want_flags)
    ts = np.concatenate(pooled[::-1])                      # any rank order: the merge sorts
    seg, res = s.merge_segments(ts, mp, True)
    assert seg.tobytes() == want_seg.tobytes()
    assert res["do_cut"] == want_res["do_cut"] and res["n_timestamps"] == want_res["n_timestamps"]
    assert bits([res["time_removed"], res["saved_pct"]]).tolist() == \
        bits([want_res["time_removed"], want_res["saved_pct"]]).tolist()
    # and identical to the whole stream scanned + merged on the device in one go
    fl_all = s.check_frames_device(d_mv, torch.from_numpy(off64).cuda(), torch.from_numpy(sd).cuda())
    dseg, dres = s.merge_streams_device(fl_all, torch.from_numpy(pts).cuda(),
                                        torch.tensor([0, n], dtype=torch.int64, device="cuda"),
                                        torch.from_numpy(mp.to_record().view(np.uint8).copy()).cuda(), True, 64)
    torch.cuda.synchronize()
    k = int(m.results_from_bytes(dres.cpu().numpy())[0]["n_segments"])
    assert dseg[0, :k].cpu().numpy().tobytes() == np.stack([want_seg["start"], want_seg["end"]], 1).tobytes()


# ------------------------------------------------------------------ f2: extract_mvs JSON -> HIP scan

def test_mvjson_roundtrip_feeds_the_hip_scan(gpu_scanner_factory, tmp_path):
    """write_json -> read_json (schema of tools/extract_mvs.cpp:97-169) -> mtgpu_scan_frames:
    same flags as the direct-record scan and as the oracle; likewise through the .mtmv container."""
    spec = synth.StreamSpec(width=640, height=480, block=16, sub=2, fps=25.0, gop=10, seed=31)
    spec.events = [synth.Event(2, 14, 10, 8, 4, 3, 9, -4), synth.Event(20, 28, 25, 15, 3, 3, -7, 2)]
    n = 30
    frames = [synth.gen_frame(spec, i) for i in range(n)]
    pts = [spec.pts_seconds(i) for i in range(n)]
    p = ob.params_from_config(640, 480)
    s = gpu_scanner_factory(m.ScanParams.from_config(640, 480))
    b0 = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b0.mv, b0.frame_off, b0.has_sd)
    assert 0 < want.sum() < n
    direct = s.check_frames(b0)
    assert np.array_equal(direct, want)
    js = str(tmp_path / "s.json")
    m.mvjson.write_json(js, frames, pts, (1, 90000))
    f2, p2, tb = m.mvjson.read_json(js)
    b2 = m.FrameBatch.from_frames(f2)
    assert np.array_equal(s.check_frames(b2), want)
    assert np.array_equal(scan_compact(s, b2.mv, b2.frame_off, b2.has_sd), want)
    # the timestamps the JSON carries (%.6f seconds) merge to the oracle's segments for those values
    mp = m.MergeParams(duration=n / 25.0, max_gap_sec=0.2, padding_sec=0.1, min_savings_pct=5.0)
    ts = [t for t, f in zip(p2, want) if f]
    seg, res = s.merge_segments(ts, mp, True)
    wseg, wres = ob.pool_and_merge(ts, mp, True)
    assert seg.tobytes() == wseg.tobytes() and res["do_cut"] == wres["do_cut"]
    mt = str(tmp_path / "s.mtmv")
    m.mvfile.write_mtmv(mt, 640, 480, 1, 90000, 25.0, n / 25.0, [spec.pts_ticks(i) for i in range(n)], frames)
    _, tab, mv = m.mvfile.read_mtmv(mt)
    b1 = m.FrameBatch.from_frames(m.mvfile.frames_of(tab, mv))
    assert np.array_equal(s.check_frames(b1), want)


# ------------------------------------------------------------------ compact records + host dispatcher

def test_compact_records_match_aos_everywhere(gpu_scanner_factory):
    """The 8-byte compact layout (what the host dispatcher ships) gives the flags of the 40-byte
    layout for every counter form, for banded and sliced plans and on ragged adversarial frames."""
    rng = np.random.RandomState(99)
    cases = [(1920, 1080, dict(), None, 0), (1920, 1080, dict(vectors_needed=3), 4, 0),
             (1920, 1080, dict(vectors_needed=9), 108, 2), (3840, 2160, dict(), None, 0),
             (3840, 2160, dict(block_size=4, block_shift=2, vectors_needed=1), None, 4),
             (3840, 2160, dict(block_size=4, block_shift=2, vectors_needed=4), None, 0),     # row bands (spill queue)
             (3840, 2160, dict(block_size=4, block_shift=2, vectors_needed=1), 32, 0),       # many row bands
             (1000, 600, dict(block_size=1, block_shift=0, vectors_needed=2, vertical_mask=0.0), None, 0)]
    for (w, h, kw, fb, slices) in cases:
        p = ob.params_from_config(w, h, **kw)
        s = gpu_scanner_factory(m.ScanParams.from_config(w, h, **kw), force_fb=fb)
        if slices:
            s.set_slices(slices)
        mv, off, sd = synth.random_frames(rng, 20, 6000, w, h, hot=0.5)
        want = ob.scan_frames(p, mv, off, sd)
        assert np.array_equal(s.check_frames(m.FrameBatch(mv, off, None, sd)), want), (w, h, kw, fb)
        assert np.array_equal(scan_compact(s, mv, off, sd), want), (w, h, kw, fb, s.plan)
        assert np.array_equal(scan_compact(s, mv, off, None), ob.scan_frames(p, mv, off, None))


def test_compact_records_any_8_byte_alignment(gpu_scanner_factory):
    """The compact array may start at any 8-byte boundary (the kernel loads record PAIRS with
    16-byte loads from the first 16-byte aligned record on and handles the odd ends separately);
    frames with 0, 1, 2 and 3 records; an odd base is rejected."""
    import torch
    rng = np.random.RandomState(123)
    p = ob.params_from_config(1920, 1080, vectors_needed=1, clusters_needed=1)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080, vectors_needed=1, clusters_needed=1))
    mv, off, sd = synth.random_frames(rng, 40, 3000, 1920, 1080, hot=0.5)
    tiny = [mv[:0], mv[5:6], mv[10:12], mv[20:23]]                 # 0, 1, 2, 3 records
    b = m.FrameBatch.from_frames([mv[int(off[i]):int(off[i + 1])] if sd[i] else None for i in range(40)] + tiny)
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    rec = torch.from_numpy(m.pack_records(b.mv).view(np.uint8).reshape(-1).copy())
    d_off = torch.from_numpy(b.frame_off.astype(np.int64)).cuda()
    d_sd = torch.from_numpy(b.has_sd).cuda()
    for shift in (0, 8, 16, 24, 40):
        buf = torch.zeros(rec.numel() + 64, dtype=torch.uint8, device="cuda")
        view = buf[shift:shift + rec.numel()]
        view.copy_(rec)
        assert view.data_ptr() % 16 == shift % 16
        for slices in (1, 2):
            s.set_slices(slices)
            got = s.check_frames_device_compact(view, d_off, d_sd).cpu().numpy()
            assert np.array_equal(got, want), (shift, slices)
    s.set_slices(0)
    with pytest.raises(m.MtgpuError) as ei:
        s.check_frames_device_compact(torch.zeros(rec.numel() + 8, dtype=torch.uint8, device="cuda")[4:4 + rec.numel()], d_off, d_sd)
    assert ei.value.code == 1


@pytest.mark.parametrize("layout", [m.LAYOUT_COMPACT8, m.LAYOUT_AOS40, m.LAYOUT_COMPACT8 | m.LAYOUT_ZERO_COPY,
                                    m.LAYOUT_AOS40 | m.LAYOUT_ZERO_COPY])
def test_scan_pipe_layouts_and_oversize_frames(gpu_scanner_factory, layout):
    """Both staging layouts of the pinned pipe against the oracle, including a frame larger than a
    whole batch (the reference's check_frame accepts any record count: the pipe grows the batch)."""
    spec = synth.spec_1080p(seed=17, sub=1)
    spec.events = synth.scripted_events(spec, 90)
    frames = [synth.gen_frame(spec, i) for i in range(90)]
    frames[7] = np.zeros(0, dtype=m.MV_DTYPE)
    big = synth.gen_frame(synth.spec_1080p(seed=18, sub=2), 5)          # 32 640 records among 8 160-record frames
    frames[40] = big
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080, vectors_needed=1))
    b = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    for (max_rec, max_fr, nbuf) in [(8160 * 5, 7, 2), (8160, 1, 1), (100, 4, 2), (8160 * 3 + 17, 1000, 4)]:
        pipe = m.ScanPipe(s, max_rec, max_fr, nbuf, layout=layout)
        for i, f in enumerate(frames):
            pipe.feed(f, spec.pts_seconds(i), tag=i)
        out = pipe.drain()
        assert [t for _, _, t in out] == list(range(90))
        assert [fl for _, fl, _ in out] == want.tolist(), (max_rec, max_fr, nbuf)
        pipe.close()


def test_pipe_submit_failure_leaves_batches_usable(gpu_scanner_factory, monkeypatch):
    """State machine under an injected failure (MTGPU_INJECT_SUBMIT_FAIL: the 2nd submit fails
    after its H2D copies were queued): the call reports the error, the batch is still being filled
    (re-submit works), nothing leaks: the pipe finishes the stream with correct flags."""
    import ctypes as C
    spec = synth.spec_1080p(seed=23, sub=1)
    spec.events = synth.scripted_events(spec, 40)
    frames = [synth.gen_frame(spec, i) for i in range(40)]
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080, vectors_needed=1))
    b = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    monkeypatch.setenv("MTGPU_INJECT_SUBMIT_FAIL", "2")
    pipe = m.ScanPipe(s, 8160 * 4, 4, 2)
    monkeypatch.delenv("MTGPU_INJECT_SUBMIT_FAIL")
    lib = m.load_library()
    failures = 0
    for i, f in enumerate(frames):
        try:
            pipe.feed(f, spec.pts_seconds(i), tag=i)
        except m.MtgpuError as e:
            assert e.code == 3 and "injected" in str(e)
            failures += 1
            # the failed batch is still in state "filling": submit it again, then feed the frame
            m._abi.check(lib.mtgpu_pipe_submit(pipe._pipe, pipe._cur))
            pipe._cur = None
            pipe._inflight += 1
            pipe.feed(f, spec.pts_seconds(i), tag=i)
    assert failures == 1
    out = pipe.drain()
    assert [t for _, _, t in out] == list(range(40)) and [fl for _, fl, _ in out] == want.tolist()
    # every staging batch is free again
    held = []
    for _ in range(2):
        h = C.c_void_p()
        m._abi.check(lib.mtgpu_pipe_acquire(pipe._pipe, C.byref(h)))
        held.append(h)
    assert lib.mtgpu_pipe_acquire(pipe._pipe, C.byref(C.c_void_p())) == m._abi.MT_ERR_BUSY
    for h in held:
        m._abi.check(lib.mtgpu_pipe_release(pipe._pipe, h))
    pipe.close()


@pytest.mark.parametrize("poison", [False, True])
def test_pipe_collect_failure_never_frees_a_running_batch(gpu_scanner_factory, monkeypatch, poison):
    """a failed wait in mtgpu_pipe_collect used to hand the batch out while its zero-copy kernel
    could still be reading the pinned staging.  Now the collect drains the batch's stream first
    (MTGPU_INJECT_COLLECT_FAIL=2: the 2nd collect's wait fails); if that drain fails as well (=-2) the batch is
    retired: release accepts it, acquire never returns it, the pipe keeps working on the remaining batch."""
    import ctypes as C
    spec = synth.spec_1080p(seed=31, sub=1)
    spec.events = synth.scripted_events(spec, 48)
    frames = [synth.gen_frame(spec, i) for i in range(48)]
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080, vectors_needed=1))
    b = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    monkeypatch.setenv("MTGPU_INJECT_COLLECT_FAIL", "-2" if poison else "2")
    pipe = m.ScanPipe(s, 8160 * 4, 4, 2)
    monkeypatch.delenv("MTGPU_INJECT_COLLECT_FAIL")
    lib = m.load_library()
    failures = 0
    for i, f in enumerate(frames):
        try:
            pipe.feed(f, spec.pts_seconds(i), tag=i)
        except m.MtgpuError as e:
            assert e.code == 3 and ("retired" in str(e) if poison else "injected" in str(e))
            failures += 1
            pipe.feed(f, spec.pts_seconds(i), tag=i)       # the wrapper gave the failed batch back: carry on
    assert failures == 1
    out = pipe.drain()
    tags = [t for _, _, t in out]
    lost = sorted(set(range(48)) - set(tags))
    assert len(lost) == 4 and lost == list(range(lost[0], lost[0] + 4))      # exactly the failed batch's frames
    assert tags == sorted(tags) and [fl for _, fl, _ in out] == want[tags].tolist()
    held = []
    for _ in range(1 if poison else 2):
        h = C.c_void_p()
        m._abi.check(lib.mtgpu_pipe_acquire(pipe._pipe, C.byref(h)))
        held.append(h)
    assert lib.mtgpu_pipe_acquire(pipe._pipe, C.byref(C.c_void_p())) == m._abi.MT_ERR_BUSY
    for h in held:
        m._abi.check(lib.mtgpu_pipe_release(pipe._pipe, h))
    st = m._abi.PipeStatsC()
    m._abi.check(lib.mtgpu_pipe_get_stats(pipe._pipe, C.byref(st)))
    assert st.n_buffers == 2 and st.device_bytes == 0 and st.pinned_bytes >= 2 * 8160 * 4 * 8
    pipe.close()


@pytest.mark.parametrize("layout", ["compact_zc", "aos_copy"])
def test_pipe_growth_failure_keeps_the_batch_usable(gpu_scanner_factory, monkeypatch, layout):
    """A frame larger than a whole batch makes the pipe grow an empty batch; when that allocation fails
    (MTGPU_INJECT_GROW_FAIL: the failure is raised inside the re-allocation, after the new pinned block
    was obtained) the call reports MT_ERR_NOMEM and the batch keeps its previous staging: frames that
    fit still go through with correct flags, and without the injection the same frame is accepted."""
    spec = synth.spec_1080p(seed=29, sub=1)
    spec.events = synth.scripted_events(spec, 40)
    frames = [synth.gen_frame(spec, i) for i in range(40)]
    big = np.concatenate([f for f in frames[1:6]])                # 5 frames' records as ONE frame: 40 800 records
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080, vectors_needed=1))
    b = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    want_big = int(ob.check_frame(p, big))
    lay = (m.LAYOUT_COMPACT8 | m.LAYOUT_ZERO_COPY) if layout == "compact_zc" else m.LAYOUT_AOS40
    monkeypatch.setenv("MTGPU_INJECT_GROW_FAIL", "1")
    pipe = m.ScanPipe(s, 8160 * 2, 4, 2, layout=lay)
    monkeypatch.delenv("MTGPU_INJECT_GROW_FAIL")
    for i in range(0, 10):
        pipe.feed(frames[i], float(i), tag=i)
    assert [fl for _, fl, _ in pipe.drain()] == want[:10].tolist()
    with pytest.raises(m.MtgpuError) as e:
        pipe.feed(big, 99.0, tag=99)                              # needs growth -> injected failure
    assert e.value.code == m._abi.MT_ERR_NOMEM and "injected" in str(e.value)
    for i in range(10, 40):                                       # the same batch, old staging intact
        pipe.feed(frames[i], float(i), tag=i)
    out = pipe.drain()
    assert [t for _, _, t in out] == list(range(10, 40)) and [fl for _, fl, _ in out] == want[10:].tolist()
    pipe.close()
    pipe = m.ScanPipe(s, 8160 * 2, 4, 2, layout=lay)              # no injection: the batch grows
    pipe.feed(frames[1], 0.0, tag=0)
    pipe.feed(big, 1.0, tag=1)
    pipe.feed(frames[2], 2.0, tag=2)
    assert [fl for _, fl, _ in pipe.drain()] == [int(want[1]), want_big, int(want[2])]
    pipe.close()


def _fine_stream_file(tmp_path, n):
    spec = synth.StreamSpec(width=3840, height=2160, block=4, sub=1, fps=30.0, gop=12, seed=61, salt_p=1e-4)
    spec.events = [synth.Event(3, 9, 300, 200, 8, 6, 9, 3), synth.Event(20, 26, 500, 100, 5, 5, -7, 0)]
    frames = [synth.gen_frame(spec, i) for i in range(n)]
    ticks = [spec.pts_ticks(i) for i in range(n)]
    path = str(tmp_path / "fine.mtmv")
    m.mvfile.write_mtmv(path, 3840, 2160, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
    return spec, frames, ticks, path


def test_cpp_host_pipeline_4k_fine_and_failure_exit_code(tmp_path):
    """mtgpu_scan_file on a 4K / 4x4-block stream (518 400 records per frame: far beyond the old
    fixed 131 072-record batch) against the oracle-driven transcription of the reference's worker
    loop; and a failing scan must exit non-zero with the error on stderr (never rc 0 with partial
    segments)."""
    from test_gpu_parity import _check_job, _reference_worker_loop
    exe = os.path.join(os.path.dirname(m.LIB_PATH), "mtgpu_scan_file")
    n = 36
    spec, frames, ticks, path = _fine_stream_file(tmp_path, n)
    cfg = dict(BLOCK_SIZE="4", BLOCK_SHIFT="2", VECTORS_NEEDED="1", CHUNK_DURATION_SEC="0.5", TARGET_FPS="0",
               MAX_GAP_SEC="0.2", PADDING_SEC="0.1", MIN_SAVINGS_PCT="5")
    env = dict(os.environ, **cfg)
    p = ob.params_from_config(3840, 2160, block_size=4, block_shift=2, vectors_needed=1)
    mp = m.MergeParams(duration=n / spec.fps, max_gap_sec=0.2, padding_sec=0.1, min_savings_pct=5.0)
    pooled, (want_seg, want_res) = _reference_worker_loop(spec, frames, ticks, n / spec.fps, p, 0.5, 0.0, mp)
    assert len(pooled) >= 8 and len(want_seg) >= 2
    for staging in ("compact8", "aos40", "compact8_zc"):
        out = subprocess.run([exe, path, "--threads", "2"], check=True, capture_output=True, text=True,
                             env=dict(env, MTGPU_STAGING=staging)).stdout
        _check_job(json.loads(out), pooled, want_seg, want_res)
    # a failing submit inside one worker: the process must fail loudly
    bad = subprocess.run([exe, path, "--threads", "2"], capture_output=True, text=True,
                         env=dict(env, MTGPU_INJECT_SUBMIT_FAIL="1"))
    assert bad.returncode != 0 and "injected" in bad.stderr and bad.stdout.strip() == ""


# ------------------------------------------------------------------ large merges (multi-workgroup path)

def _merge_inputs(rng, n, variant):
    dur = 86400.0
    base = np.sort(rng.rand(n) * dur)
    ts = np.round(base / 60.0) * 60.0 + rng.rand(n) * 20.0       # clusters: gaps on both sides of MAX_GAP_SEC
    if variant == "sorted":
        return np.sort(ts)
    if variant == "chunk_runs":                                  # pooled per-chunk results: sorted runs, any order
        v = np.sort(ts)
        runs = np.array_split(v, max(1, n // 700))
        order = rng.permutation(len(runs))
        return np.concatenate([runs[i] for i in order])
    if variant == "dups":
        return rng.permutation(np.concatenate([ts, ts[: n // 2 + 1], ts[:1]]))
    return rng.permutation(ts)


def _check_merge_equal(got_seg, got_res, want_seg, want_res):
    assert got_res["n_timestamps"] == want_res["n_timestamps"]
    assert got_res["n_segments"] == want_res["n_segments"] and got_res["do_cut"] == want_res["do_cut"]
    assert bits([got_res["time_removed"], got_res["saved_pct"]]).tolist() == \
        bits([want_res["time_removed"], want_res["saved_pct"]]).tolist()
    assert np.array_equal(bits(got_seg["start"]), bits(want_seg["start"]))
    assert np.array_equal(bits(got_seg["end"]), bits(want_seg["end"]))


@pytest.mark.parametrize("job", [False, True])
def test_merge_large_path_edge_sizes(gpu_scanner_factory, monkeypatch, job):
    """The multi-workgroup merge (tile sort + merge-path passes + scan) forced on for every size
    (MTGPU_MERGE_LARGE_MIN=1): tile / run boundaries, duplicates, all-equal, every gap a split."""
    monkeypatch.setenv("MTGPU_MERGE_LARGE_MIN", "1")
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080))
    monkeypatch.delenv("MTGPU_MERGE_LARGE_MIN")
    rng = np.random.RandomState(11)
    mps = [m.MergeParams(duration=86400.0, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0),
           m.MergeParams(duration=40000.0, max_gap_sec=0.0, padding_sec=2.0, min_savings_pct=5.0),
           m.MergeParams(duration=0.0, max_gap_sec=-1.0, padding_sec=0.0, min_savings_pct=5.0)]
    for n in [1, 2, 3, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 8192, 8193, 20000, 70000]:
        for variant in ("sorted", "shuffled", "dups", "chunk_runs"):
            v = _merge_inputs(rng, n, variant)
            for mp in mps:
                want = ob.pool_and_merge(v, mp, job)
                got = s.merge_segments(v, mp, job)
                _check_merge_equal(got[0], got[1], want[0], want[1])
    same = np.full(5000, 7.25)
    for mp in mps:
        _check_merge_equal(*s.merge_segments(same, mp, job), *ob.pool_and_merge(same, mp, job))
    # the edges of binary64 through the multi-workgroup path: denormals, 1e308, negatives, 2^53 neighbours
    extreme = np.r_[5e-324, 1e-323, 2.2250738585072014e-308, -1e308, -3.5, -0.0, 1e308, 1.7976931348623157e308,
                    2.0 ** 53, 2.0 ** 53 + 2, rng.rand(4200) * 1e4, np.arange(300) * 1e-310]
    for mp in mps + [m.MergeParams(duration=1e308, max_gap_sec=1e-320, padding_sec=5e-324, min_savings_pct=0.0)]:
        v = rng.permutation(extreme)
        _check_merge_equal(*s.merge_segments(v, mp, job), *ob.pool_and_merge(v, mp, job))
    with pytest.raises(m.MtgpuError) as ei:
        s.merge_segments(np.r_[rng.rand(3000), float("nan"), rng.rand(3000)], mps[0], job)
    assert ei.value.code == 1
    # NaN through the device entry point: the same result record from the multi-workgroup path (forced on here)
    # and from the one-workgroup kernel (a fresh context without the override) — ADVICE r2
    import torch
    bad = torch.tensor([3.0, float("nan"), 1.0, 2.0], dtype=torch.float64, device="cuda:0")
    recs = []
    for scanner in (s, gpu_scanner_factory(s.params)):
        _, res = scanner.merge_timestamps_device(bad, mps[0], job)
        torch.cuda.synchronize()
        recs.append(m.results_from_bytes(res.cpu().numpy().reshape(1, -1))[0])
    assert recs[0].tobytes() == recs[1].tobytes()
    assert recs[0]["status"] == 1 and recs[0]["n_timestamps"] == 0 and recs[0]["do_cut"] == -1
    with pytest.raises(m.MtgpuError) as ei:                      # capacity: need reported, like the small path
        s.merge_segments(np.arange(6000.0) * 10.0, mps[0], False, cap=10)
    assert ei.value.code == 2


@pytest.mark.parametrize("n", [100_000, 1_000_000])
def test_merge_large_inputs_bit_exact(gpu_scanner_factory, n):
    """Pooled timestamp lists of 10^5 / 10^6 entries (a day of footage at ~10 fps): shuffled, with
    duplicates, and as out-of-order sorted runs — bit-exact against the oracle, through the host
    entry point and the device-resident one."""
    import torch
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080))
    rng = np.random.RandomState(n % 1000 + 3)
    mp = m.MergeParams(duration=86400.0, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0)
    for variant in ("shuffled", "dups", "chunk_runs"):
        v = _merge_inputs(rng, n, variant)
        want_seg, want_res = ob.pool_and_merge(v, mp, True)
        assert want_res["n_segments"] > 100
        got_seg, got_res = s.merge_segments(v, mp, True)
        _check_merge_equal(got_seg, got_res, want_seg, want_res)
        cap = int(want_res["n_segments"]) + 8
        dseg, dres = s.merge_timestamps_device(torch.from_numpy(v).cuda(), mp, True, seg_cap=cap)
        torch.cuda.synchronize()
        rec = m.results_from_bytes(dres.cpu().numpy()[None, :])[0]
        k = int(rec["n_segments"])
        assert k == want_res["n_segments"] and int(rec["n_timestamps"]) == want_res["n_timestamps"]
        assert dseg[:k].cpu().numpy().tobytes() == np.stack([want_seg["start"], want_seg["end"]], 1).tobytes()
        assert bits([rec["time_removed"], rec["saved_pct"]]).tolist() == \
            bits([want_res["time_removed"], want_res["saved_pct"]]).tolist()
    # worst case for the in-order sum: every timestamp its own segment
    v = rng.permutation(np.arange(n, dtype=np.float64) * 7.0)
    _check_merge_equal(*s.merge_segments(v, mp, False), *ob.pool_and_merge(v, mp, False))


# ------------------------------------------------------------------ parameter sets as the reference parses them

def _varied_stream(rng, width, height, n_frames):
    """8x8-block frames with jitter of 0-1 px and a few moving rectangles of assorted speed, size and
    height, so that threshold, votes-per-cell, cluster count and vertical mask each flip some frames."""
    bx, by = width // 8, height // 8
    gx, gy = np.meshgrid(np.arange(bx), np.arange(by))
    frames = []
    for f in range(n_frames):
        if f % 12 == 0:
            frames.append(None)                          # I-frame: no side data
            continue
        dx = rng.integers(-1, 2, size=(by, bx))
        dy = rng.integers(-1, 2, size=(by, bx)) * (rng.random((by, bx)) < 0.3)
        for _ in range(int(rng.integers(0, 4))):
            x0, y0 = int(rng.integers(0, bx - 2)), int(rng.integers(0, by - 2))
            w, h = int(rng.integers(1, 9)), int(rng.integers(1, 7))
            sp = int(rng.choice([2, 3, 4, 5, 9, 14]))
            sel = (gx >= x0) & (gx < x0 + w) & (gy >= y0) & (gy < y0 + h)
            dx = np.where(sel, sp, dx)
            dy = np.where(sel, int(rng.integers(-2, 3)), dy)
        keep = rng.random((by, bx)) < 0.9                # a few blocks carry no vector
        rec = np.zeros(int(keep.sum()), dtype=m.MV_DTYPE)
        rec["dst_x"] = (4 + 8 * gx)[keep]
        rec["dst_y"] = (4 + 8 * gy)[keep]
        rec["src_x"] = rec["dst_x"] - dx[keep]
        rec["src_y"] = rec["dst_y"] - dy[keep]
        rec["w"] = rec["h"] = 8
        rec["source"] = -1
        frames.append(rec)
    b = m.FrameBatch.from_frames(frames)
    return b.mv, b.frame_off, b.has_sd


def test_scan_under_reference_parsed_configs(gpu_scanner_factory):
    """Every environment of tests/golden/reference_host_vectors.json (answers of the reference's own
    config.hpp, compiled and run: prefix parses, hex, the uint8 wrap of VECTORS_NEEDED, odd masks and
    thresholds) becomes a parameter set through mtgpu_params_from_config and scans one 640x368 stream on
    the HIP path, 40-byte and compact records, against the oracle under the same values."""
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_host_vectors.json")))
    mv, off, has_sd = _varied_stream(np.random.default_rng(31), 640, 368, 48)
    seen, ran, some_motion = set(), 0, 0
    for case in vec["config"]:
        ans = {ln.split()[0]: ln.split() for ln in case["answers"]}
        keys = ("mv_threshold_sq", "block_size", "block_shift", "vectors_needed", "clusters_needed", "vertical_mask")
        if any(ans[k][1] == "error" for k in keys):
            continue                                     # the reference terminates on these (uncaught std exception)
        vals = (float(ans["mv_threshold_sq"][2]), int(ans["block_size"][2]), int(ans["block_shift"][2]),
                int(ans["vectors_needed"][2]), int(ans["clusters_needed"][2]), float(ans["vertical_mask"][2]))
        key = tuple(repr(v) for v in vals)
        if key in seen:
            continue
        seen.add(key)
        try:
            po = ob.params_from_config(640, 368, *vals)
        except Exception:
            po = None
        try:
            p = m.ScanParams.from_config(640, 368, *vals)
        except m.MtgpuError:
            assert po is None, f"product rejects {case['name']} but the oracle accepts it"
            continue
        assert po is not None, f"oracle rejects {case['name']} but the product accepts it"
        try:
            want = ob.scan_frames(po, mv, off, has_sd)
        except Exception:
            with pytest.raises(m.MtgpuError):            # outside the defined domain for both
                gpu_scanner_factory(p)
            continue
        s = gpu_scanner_factory(p)
        b = m.FrameBatch(mv, off, None, has_sd)
        assert np.array_equal(s.check_frames(b), want), case["name"]
        assert np.array_equal(scan_compact(s, mv, off, has_sd), want), case["name"]
        ran += 1
        some_motion += int(want.any())
        s.close()
    assert ran >= 25 and some_motion >= 5, (ran, some_motion)


@pytest.mark.parametrize("layout", [m._abi.LAYOUT_COMPACT8 | m._abi.LAYOUT_ZERO_COPY, m._abi.LAYOUT_AOS40])
def test_pipe_pins_staging_on_first_use(gpu_scanner_factory, monkeypatch, layout):
    """Round 4: page-locking is the cost of creating a pipe (~0.2 ms per MiB, serialised across threads by the
    driver), so a pipe pins ONE batch at creation and every other batch when mtgpu_pipe_acquire first hands it
    out; MTGPU_PIPE_EAGER=1 restores pin-everything-at-creation.  Results do not depend on it."""
    import ctypes as C
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    lib = m.load_library()

    def stats(pipe):
        st = m._abi.PipeStatsC()
        m._abi.check(lib.mtgpu_pipe_get_stats(pipe._pipe, C.byref(st)))
        return st
    spec = synth.spec_1080p(seed=12, sub=1)
    spec.events = [synth.Event(2, 20, 40, 30, 4, 3, 9, 2)]
    mv, off, pts, sd = synth.gen_stream(spec, 24)
    want = ob.scan_frames(p, mv, off, sd)
    pipe = m.ScanPipe(s, 8160 * 4, 4, 3, layout=layout)
    st = stats(pipe)
    one = st.pinned_bytes
    assert st.pinned_batches == 1 and st.n_buffers == 3 and one >= 8160 * 4 * (8 if layout & 1 == 0 else 40)
    assert (st.device_bytes == 0) == bool(layout & m._abi.LAYOUT_ZERO_COPY)
    a, b = C.c_void_p(), C.c_void_p()
    m._abi.check(lib.mtgpu_pipe_acquire(pipe._pipe, C.byref(a)))          # the pinned one
    assert stats(pipe).pinned_batches == 1
    m._abi.check(lib.mtgpu_pipe_acquire(pipe._pipe, C.byref(b)))          # a second one while the first is held: pinned now
    st = stats(pipe)
    assert st.pinned_batches == 2 and st.pinned_bytes == 2 * one and st.pin_us > 0
    m._abi.check(lib.mtgpu_pipe_release(pipe._pipe, a))
    m._abi.check(lib.mtgpu_pipe_release(pipe._pipe, b))
    # a whole stream through it (4 frames per batch -> up to 3 batches in flight -> all three get pinned)
    for i in range(24):
        fr = mv[int(off[i]):int(off[i + 1])]
        pipe.feed(fr if sd[i] else None, float(pts[i]), tag=i)
    got = pipe.drain()
    assert [f for _, f, _ in got] == want.tolist() and want.sum() >= 10
    st = stats(pipe)
    assert 2 <= st.pinned_batches <= 3 and st.pinned_bytes == st.pinned_batches * one
    pipe.close()
    if not experiments_build():                  # MTGPU_PIPE_EAGER is an A/B knob: read by the experiments build only
        return
    monkeypatch.setenv("MTGPU_PIPE_EAGER", "1")
    pipe = m.ScanPipe(s, 8160 * 4, 4, 3, layout=layout)
    st = stats(pipe)
    assert st.pinned_batches == 3 and st.pinned_bytes == 3 * one
    for i in range(24):
        fr = mv[int(off[i]):int(off[i + 1])]
        pipe.feed(fr if sd[i] else None, float(pts[i]), tag=i)
    assert [f for _, f, _ in pipe.drain()] == want.tolist()
    pipe.close()


def test_pipe_batches_run_on_the_contexts_stream_pool(gpu_scanner_factory, monkeypatch):
    """Round 4: a pipe no longer creates a HIP stream per batch (3.5 ms each, serialised by the runtime — 192 of
    them were most of the 0.6 s a worker spent in mtgpu_pipe_create at 64 x 1); its batches take streams from a
    pool of 8 owned by the context.  Four pipes of 3 batches on one context: 8 pool streams + the context's own,
    none owned by a pipe, results as before; MTGPU_PIPE_STREAMS=0 (read at mtgpu_create) gives every batch its
    own stream again."""
    import ctypes as C
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    lib = m.load_library()
    spec = synth.spec_1080p(seed=13, sub=1)
    spec.events = [synth.Event(3, 15, 40, 30, 4, 3, 9, 2)]
    mv, off, pts, sd = synth.gen_stream(spec, 20)
    want = ob.scan_frames(p, mv, off, sd).tolist()

    def pipe_stats(pipe):
        st = m._abi.PipeStatsC()
        m._abi.check(lib.mtgpu_pipe_get_stats(pipe._pipe, C.byref(st)))
        return st
    for pooled in ((True, False) if experiments_build() else (True,)):     # MTGPU_PIPE_STREAMS: experiments build only
        if not pooled:
            monkeypatch.setenv("MTGPU_PIPE_STREAMS", "0")
        s = gpu_scanner_factory(p)
        pipes = [m.ScanPipe(s, 8160 * 3, 3, 3) for _ in range(4)]
        for i in range(20):                                     # four interleaved "decoder threads"
            fr = mv[int(off[i]):int(off[i + 1])]
            for pp in pipes:
                pp.feed(fr if sd[i] else None, float(pts[i]), tag=i)
        for pp in pipes:
            assert [f for _, f, _ in pp.drain()] == want
        assert [pipe_stats(pp).hip_streams for pp in pipes] == ([0] * 4 if pooled else [3] * 4)
        assert s.stats()["hip_streams"] == (9 if pooled else 1)
        for pp in pipes:
            pp.close()


@pytest.mark.parametrize("layout", [m._abi.LAYOUT_COMPACT8 | m._abi.LAYOUT_ZERO_COPY, m._abi.LAYOUT_AOS40 | m._abi.LAYOUT_ZERO_COPY,
                                    m._abi.LAYOUT_COMPACT8])
def test_pipe_staging_reuse_stress(gpu_scanner_factory, layout):
    """The same three staging blocks refilled ~1500 times with different frames, every flag checked: what the host
    writes into a block (non-temporal stores for compact staging) must be what the NEXT kernel on that block reads,
    and what that kernel writes (flag bytes, straight into pinned memory with zero-copy) what the host reads after
    the batch's event — no stale line from the block's previous use on either side.  (Round 4: registered
    user-pointer staging failed exactly this, once, in the soak; driver-allocated pinned staging must never.)"""
    p = ob.params_from_config(640, 480, vectors_needed=1, clusters_needed=1, mv_threshold_sq=4.0)
    s = gpu_scanner_factory(p)
    rng = np.random.RandomState(layout + 5)
    # a pool of small frames with known answers, half of them "motion"
    pool = []
    for k in range(64):
        n = int(rng.randint(1, 400))
        mv = np.zeros(n, dtype=m.MV_DTYPE)
        mv["dst_x"], mv["dst_y"] = rng.randint(0, 640, size=n), rng.randint(32, 448, size=n)
        mv["src_x"], mv["src_y"] = mv["dst_x"] - 1, mv["dst_y"]                       # below the threshold
        if k & 1:                                                                     # two neighbouring cells with a real vector
            gx, gy = int(rng.randint(2, 36)), int(rng.randint(3, 26))
            for j, (cx, cy) in enumerate(((gx, gy), (gx + 1, gy))):
                i = int(rng.randint(0, n)) if n > 2 else j % n
                mv["dst_x"][i], mv["dst_y"][i] = cx * 16 + 5, cy * 16 + 5
                mv["src_x"][i], mv["src_y"][i] = cx * 16 + 5 - 9, cy * 16 + 5
        pool.append(mv)
    b = m.FrameBatch.from_frames(pool)
    want_pool = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    assert 10 < want_pool.sum() < 54
    pipe = m.ScanPipe(s, 4 * 400, 4, 3, layout=layout)
    order = rng.randint(0, 64, size=6000)
    for t, k in enumerate(order):
        pipe.feed(pool[k], float(t), tag=int(k))
    got = pipe.drain()
    pipe.close()
    assert len(got) == 6000
    bad = [(i, tag) for i, (_, fl, tag) in enumerate(got) if fl != want_pool[tag]]
    assert not bad, bad[:10]


def test_pipe_flag_bytes_own_their_lines(gpu_scanner_factory):
    """Round 5: the flag bytes — the only part of a staging block the DEVICE writes — start on a 128-byte
    line of their own and no host-written array (pts, tags) shares a line with them, for every frame capacity that
    used to put them right behind the tags; results as the oracle says."""
    import ctypes as C
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    lib = m.load_library()
    spec = synth.spec_1080p(seed=21, sub=1)
    spec.events = [synth.Event(1, 9, 40, 30, 4, 3, 9, 2)]
    mv, off, pts, sd = synth.gen_stream(spec, 12)
    want = ob.scan_frames(p, mv, off, sd).tolist()
    for max_frames in (1, 7, 12, 32, 33, 100):
        for layout in (m._abi.LAYOUT_COMPACT8 | m._abi.LAYOUT_ZERO_COPY, m._abi.LAYOUT_AOS40):
            pipe = m.ScanPipe(s, 8160 * 12, max_frames, 2, layout=layout)
            n = min(12, max_frames)
            for i in range(n):
                fr = mv[int(off[i]):int(off[i + 1])]
                pipe.feed(fr if sd[i] else None, float(pts[i]), tag=i)
            pipe._submit()
            b, fl, pt, tg, cnt = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
            m._abi.check(lib.mtgpu_pipe_collect(pipe._pipe, C.byref(b), C.byref(fl), C.byref(pt), C.byref(tg), C.byref(cnt)))
            assert cnt.value == n and fl.value % 128 == 0
            assert fl.value - (tg.value + 8 * (max_frames + 1)) >= 0           # behind the whole tag array ...
            assert (tg.value + 8 * (max_frames + 1) - 1) // 128 < fl.value // 128   # ... and on a later line than its last byte
            got = np.ctypeslib.as_array(C.cast(fl, C.POINTER(C.c_uint8)), (n,)).tolist()
            assert got == want[:n], (max_frames, layout)
            m._abi.check(lib.mtgpu_pipe_release(pipe._pipe, b))
            pipe._inflight -= 1
            pipe.close()
