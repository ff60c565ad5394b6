"""CPU tests of the oracle (oracle/mt_oracle.c) against the committed golden vectors:
hand-derived known answers, the two segments recorded from the reference's own object
code in SURVEY.md §8c, and a second independent numpy restatement.  No GPU needed."""
import json
import os

import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import synth

import oracle_binding as ob
from np_model import check_frame_np

from golden_cases import (GOLD, build_mvs, filter_case_ticks, load_filter_cases, load_hand_cases,
                          load_merge_cases, load_survey_segments, merge_case_ts)


def test_hand_case_params():
    g, _ = load_hand_cases()
    p = ob.params_from_config(**g["base"])
    for k, v in g["expect_params"].items():
        assert getattr(p, k) == v


@pytest.mark.parametrize("name,kw,case", load_hand_cases()[1], ids=lambda x: x if isinstance(x, str) else None)
def test_check_frame_hand_cases(name, kw, case):
    p = ob.params_from_config(**kw)
    mv = build_mvs(case)
    sd = bool(case.get("has_sd", 1))
    assert ob.check_frame(p, mv, sd) == case["expect"]
    flag, centres, grid = ob.check_frame(p, mv, sd, count_centres=True)
    assert (flag, centres) == (case["expect"], case["centres"])
    assert check_frame_np(p, mv, sd) == (case["expect"], case["centres"])
    # record order must not matter
    if len(mv) > 1:
        assert ob.check_frame(p, mv[::-1].copy(), sd) == case["expect"]


def test_params_derivation_known_answers():
    # SURVEY.md §8 table (reference src/motion_scanner.cpp:190-196)
    for (w, h, kw), want in [((1920, 1080, {}), (120, 68, 3)), ((3840, 2160, {}), (240, 135, 6)),
                             ((3840, 2160, dict(block_size=4, block_shift=2)), (960, 540, 27)),
                             ((1280, 720, {}), (80, 45, 2)), ((16, 16, {}), (1, 1, 0))]:
        p = ob.params_from_config(w, h, **kw)
        assert (p.grid_w, p.grid_h, p.vertical_margin) == want
    # :196 evaluates int16 * float in float32: 10 * 0.7f rounds to exactly 7.0f -> 7
    # (in double, 10 * (double)0.7f = 6.99999988 -> 6)
    assert np.float32(10) * np.float32(0.7) == np.float32(7.0)
    assert int(10 * float(np.float32(0.7))) == 6
    assert ob.params_from_config(160, 160, vertical_mask=0.7).vertical_margin == 7
    # config.hpp:75 — VECTORS_NEEDED is cast to uint8
    assert ob.params_from_config(160, 160, vectors_needed=256 + 7).vectors_needed == 7
    assert ob.params_from_config(160, 160, vectors_needed=-1).vectors_needed == 255
    # BLOCK_SIZE and BLOCK_SHIFT are independent knobs (:190-193 vs :255-256)
    p = ob.params_from_config(1920, 1080, block_size=16, block_shift=5)
    assert (p.grid_w, p.grid_h) == (60, 34)
    with pytest.raises(ValueError):
        ob.params_from_config(1920, 1080, block_shift=32)
    with pytest.raises(ValueError):
        ob.params_from_config(0, 0, block_size=0)


@pytest.mark.parametrize("seed", range(6))
def test_oracle_vs_numpy_model_random(seed):
    rng = np.random.RandomState(seed)
    cfgs = [dict(), dict(vertical_mask=0.0), dict(vectors_needed=1, clusters_needed=1),
            dict(vectors_needed=0), dict(mv_threshold_sq=float("nan")), dict(mv_threshold_sq=30.5),
            dict(clusters_needed=-3), dict(vectors_needed=200), dict(block_size=8, block_shift=3)]
    for (w, h) in [(320, 240), (1920, 1080), (48, 48), (1040, 64)]:
        for kw in cfgs:
            p = ob.params_from_config(w, h, **kw)
            mv, off, sd = synth.random_frames(rng, 6, 1500, w, h)
            flags = ob.scan_frames(p, mv, off, sd)
            for f in range(6):
                fr = mv[int(off[f]):int(off[f + 1])]
                flag, centres, _ = ob.check_frame(p, fr, bool(sd[f]), count_centres=True)
                assert (flag, centres) == check_frame_np(p, fr, bool(sd[f])), (w, h, kw, f)
                assert flags[f] == flag          # early exit == full count vs max(1, clusters_needed)


@pytest.mark.parametrize("vec", [1, 2, 4, 8, 40])
def test_oracle_vs_numpy_model_on_run_structured_frames(vec):
    """The frames the GPU's run-aggregated vote path is tested on (tests/run_frames.py: every cell's votes arrive as
    runs of 1..130 same-cell records, totals one short of / at / above VECTORS_NEEDED): the C oracle and the
    independent numpy model agree on flag AND centre count, so the checker itself is cross-checked on this input."""
    from run_frames import run_frames
    p = ob.params_from_config(1920, 1080, vectors_needed=vec, clusters_needed=2, mv_threshold_sq=4.0)
    rng = np.random.RandomState(500 + vec)
    frames = run_frames(rng, 1920, 1080, 4, vec, 24, 3)
    yes = 0
    for fr in frames:
        flag, centres, _ = ob.check_frame(p, fr, True, count_centres=True)
        assert (flag, centres) == check_frame_np(p, fr, True)
        yes += flag
    assert 0 < yes < len(frames)


def test_oracle_rejects_negative_margin():
    """vertical_margin < 0 would index before the grid in the reference (:237, :285): the oracle
    refuses it, like the product's validate_params."""
    import ctypes as C
    p = ob.params_from_config(160, 160)
    c = p.to_c()
    c.vertical_margin = -1
    grid = np.zeros(100, dtype=np.uint8)
    mv = np.zeros(1, dtype=m.MV_DTYPE)
    assert ob.lib().mto_check_frame(C.byref(c), mv.ctypes.data_as(C.c_void_p), 1, 1, grid.ctypes.data_as(C.c_void_p)) < 0
    with pytest.raises(ValueError):
        ob.scan_frames(m.ScanParams.from_c(c), mv, np.array([0, 1], dtype=np.uint64))


def test_scan_frames_threads_and_null_has_sd():
    spec = synth.spec_1080p(seed=4, sub=1)
    spec.events = synth.scripted_events(spec, 40)
    mv, off, pts, sd = synth.gen_stream(spec, 40)
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    a = ob.scan_frames(p, mv, off, sd)
    assert 0 < a.sum() < 40 and a[0] == 0 and a[30] == 0      # I-frames: no side data
    for nt in (2, 3, 8, 64):
        assert np.array_equal(ob.scan_frames(p, mv, off, sd, nthreads=nt), a)
    assert np.array_equal(ob.scan_frames(p, mv, off, None), a)


# ------------------------------------------------------------------ merge

@pytest.mark.parametrize("name,kw,case", load_merge_cases(), ids=lambda x: x if isinstance(x, str) else None)
def test_merge_hand_cases(name, kw, case):
    ts = merge_case_ts(case)
    mp = m.MergeParams(**kw)
    seg, res = ob.pool_and_merge(ts, mp, False)
    assert [list(x) for x in seg.tolist()] == case["segments"]
    assert res["time_removed"] == case["time_removed"] and res["do_cut"] == case["do_cut"]
    if "n_unique" in case:
        assert res["n_timestamps"] == case["n_unique"]
    job, jres = ob.pool_and_merge(ts, mp, True)
    assert [list(x) for x in job.tolist()] == case["job"] and jres["n_segments"] == len(case["job"])
    if kw["duration"] > 0 and ts:
        assert res["saved_pct"] == pytest.approx(case["time_removed"] / kw["duration"] * 100.0, abs=1e-9)


def test_merge_reproduces_reference_run_recorded_in_survey():
    g, ts, mp = load_survey_segments()
    seg, res = ob.pool_and_merge(ts[::-1], mp, True)
    assert res["do_cut"] == 1
    got = [["%.17g" % s, "%.17g" % e] for s, e in seg.tolist()]
    assert got == g["segments_printed_17g"]


def test_sort_unique_and_merge_input_checks():
    assert ob.sort_unique([3.0, 1.0, 2.0, 1.0, 3.0]).tolist() == [1.0, 2.0, 3.0]
    assert ob.sort_unique([]).size == 0
    assert ob.sort_unique([-0.0, 0.0]).size == 1                    # std::unique uses ==
    with pytest.raises(ValueError):
        ob.sort_unique([1.0, float("nan")])
    with pytest.raises(ValueError):
        ob.merge_segments([2.0, 1.0], m.MergeParams(duration=10.0))   # must be sorted
    with pytest.raises(ValueError):
        ob.merge_segments([1.0, 1.0], m.MergeParams(duration=10.0))   # and unique


def test_merge_sequential_python_model():
    """The oracle's single pass vs a literal two-loop python transcription of the reference's
    structure (push all segments, then clamp+sum) on random inputs: bit-equal."""
    rng = np.random.RandomState(1)
    for _ in range(200):
        n = rng.randint(1, 60)
        ts = np.unique(np.round(rng.rand(n) * 50, 3))
        mp = m.MergeParams(duration=float(rng.choice([0.0, 20.0, 50.0, 80.0])), max_gap_sec=float(rng.rand() * 4),
                           padding_sec=float(rng.rand() * 3), min_savings_pct=5.0)
        segs, cur, last = [], ts[0], ts[0]
        for t in ts[1:]:
            if t - last > mp.max_gap_sec:
                segs.append([max(0.0, cur - mp.padding_sec), last + mp.padding_sec])
                cur = t
            last = t
        segs.append([max(0.0, cur - mp.padding_sec), last + mp.padding_sec])
        out = 0.0
        for s in segs:
            s[1] = min(s[1], mp.duration)
            s[0] = min(s[0], s[1])
            out += s[1] - s[0]
        removed = mp.duration - out
        seg, res = ob.merge_segments(ts, mp, False)
        assert seg.tolist() == [tuple(s) for s in segs] or [list(x) for x in seg.tolist()] == segs
        assert res["time_removed"] == removed


# ------------------------------------------------------------------ a6 / a7

def test_frame_filter_and_chunks():
    assert ob.lib().mto_frame_skip(30.0, 10.0) == 3 and ob.lib().mto_frame_skip(30.0, 0.0) == 1
    assert ob.lib().mto_frame_skip(30.0, 30.0) == 1 and ob.lib().mto_frame_skip(25.0, 10.0) == 2
    assert ob.lib().mto_frame_skip(30.0, 45.0) == 1
    ticks = [3000 * i for i in range(300)]                        # 10 s at 30 fps, time_base 1/90000
    tb = 1.0 / 90000.0
    idx, pts = ob.filter_frames(ticks, tb, 0.0, 5.0, 1)
    assert idx == list(range(150)) and pts[1] == 3000.0 * tb
    # skip 3: the counter counts EVERY decoded frame of the call, 3rd, 6th, ... are analysed (:357)
    idx, _ = ob.filter_frames(ticks, tb, 0.0, 1.0, 3)
    assert idx == [2, 5, 8, 11, 14, 17, 20, 23, 26, 29]
    # pre-roll after a backward seek is counted by the skip counter but dropped by pts < start (:364)
    idx, _ = ob.filter_frames(ticks[50:], tb, 2.0, 3.0, 3)
    assert [i + 50 for i in idx] == [61, 64, 67, 70, 73, 76, 79, 82, 85, 88]   # local 3rd,6th.. with pts in [2,3)
    # chunks (pipeline.cpp:163-167)
    assert ob.chunks(70.0, 30.0) == [(0.0, 30.0, 0), (30.0, 60.0, 1), (60.0, 70.0, 2)]
    assert ob.chunks(60.0, 30.0) == [(0.0, 30.0, 0), (30.0, 60.0, 1)]
    assert ob.chunks(0.0, 30.0) == []


def test_frame_filter_hand_cases():
    """tests/golden/frame_filter_hand_cases.json (hand-derived from src/motion_scanner.cpp:307-314,
    357-371 and src/pipeline.cpp:163-167): the oracle AND the Python host mirror reproduce them."""
    g = load_filter_cases()
    tb = 1.0 / g["time_base_den"]
    for c in g["frame_skip"]:
        assert ob.lib().mto_frame_skip(c["fps"], c["target"]) == c["skip"], c
        assert m.frame_skip(c["fps"], c["target"]) == c["skip"], c
    for c in g["filter"]:
        ticks = filter_case_ticks(g, c)
        want_idx = [f - c["first"] for f in c["analysed"]]
        want_pts = [f / 32.0 for f in c["analysed"]]
        assert ob.filter_frames(ticks, tb, c["start"], c["end"], c["skip"]) == (want_idx, want_pts), c["name"]
        assert m.filter_frames(ticks, tb, c["start"], c["end"], c["skip"]) == (want_idx, want_pts), c["name"]
    for c in g["chunks"]:
        want = [(a, b, i) for i, (a, b) in enumerate(c["chunks"])]
        assert ob.chunks(c["duration"], c["chunk"]) == want, c
        assert m.make_chunks(c["duration"], c["chunk"]) == want, c
    # the whole-video case: chunks -> backward seek to the last keyframe -> filter
    pl = g["pipeline"]
    ticks = [g["ticks_per_frame"] * i for i in range(pl["n_frames"])]
    skip = m.frame_skip(pl["fps"], pl["target_fps"])
    pooled = []
    for k, (c0, c1, _) in enumerate(m.make_chunks(pl["duration"], pl["chunk_sec"])):
        target = int(c0 / tb)                                        # motion_scanner.cpp:322
        first = max(f for f in pl["keyframes"] if ticks[f] <= target) if c0 > 0 else 0
        for fn in (ob.filter_frames, m.filter_frames):
            idx, pts = fn(ticks[first:], tb, c0, c1, skip)
            assert [first + i for i in idx] == pl["per_chunk"][k], (k, fn)
        pooled += [first + i for i in idx if (first + i) not in pl["keyframes"]]
    assert pooled == pl["timestamps_frames"]


def test_reference_documented_examples():
    """tests/golden/reference_documented_examples.json: the behaviour the reference documents in its
    own config/motion_trim.env (threshold = "detects movement >= N pixels", cluster = "active block
    with an adjacent active neighbour", VERTICAL_MASK percentages, TARGET_FPS, gap / padding /
    savings) holds for the oracle."""
    import documented_examples as de

    def check_frames(p, frames):
        b = m.FrameBatch.from_frames(frames)
        return ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    de.run(lambda w, h, **kw: ob.params_from_config(w, h, **kw), check_frames,
           lambda ts, mp, job: ob.pool_and_merge(ts, mp, job), lambda fps, t: ob.lib().mto_frame_skip(fps, t))


# ------------------------------------------------------------------ property-based cross-check

def test_oracle_vs_numpy_model_hypothesis():
    """hypothesis-driven: arbitrary small frames (any int16 coordinates within the defined
    domain |dst - src| <= 32767), arbitrary parameters -> the C oracle (early exit) and the
    numpy model (full count) agree on the flag AND the centre count."""
    from hypothesis import given, settings, strategies as st

    coord = st.integers(-32768, 32767)

    @st.composite
    def frames(draw):
        n = draw(st.integers(0, 60))
        rows = []
        for _ in range(n):
            dx, dy = draw(st.integers(-300, 300)), draw(st.integers(-300, 300))
            x, y = draw(st.one_of(coord, st.integers(-40, 400))), draw(st.one_of(coord, st.integers(-40, 300)))
            sx, sy = max(-32768, min(32767, x - dx)), max(-32768, min(32767, y - dy))
            rows.append((sx, sy, x, y))
        return rows

    @settings(max_examples=300, deadline=None)
    @given(frames(), st.integers(8, 360), st.integers(8, 280), st.integers(0, 5),
           st.sampled_from([0.0, 1.0, 16.0, 16.5, 1e5, float("nan")]), st.integers(0, 6), st.integers(-1, 5),
           st.sampled_from([0.0, 0.05, 0.2, 0.5]), st.booleans())
    def check(rows, w, h, shift, thr, vec, clus, mask, has_sd):
        p = ob.params_from_config(w, h, mv_threshold_sq=thr, block_size=1 << shift, block_shift=shift,
                                  vectors_needed=vec, clusters_needed=clus, vertical_mask=mask)
        mv = np.zeros(len(rows), dtype=m.MV_DTYPE)
        if rows:
            a = np.array(rows, dtype=np.int64)
            mv["src_x"], mv["src_y"], mv["dst_x"], mv["dst_y"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
        flag, centres, _ = ob.check_frame(p, mv, has_sd, count_centres=True)
        assert (flag, centres) == check_frame_np(p, mv, has_sd)
        assert ob.check_frame(p, mv, has_sd) == flag

    check()


def test_bench_scan_is_the_same_scan():
    """mto_bench_scan (bench.py's timed CPU baseline: thread-local copies, passes between barriers) returns the
    flags of mto_scan_frames for every thread count, with and without has_sd, on ragged frames."""
    rng = np.random.RandomState(12)
    mv, off, sd = synth.random_frames(rng, 37, 500, 640, 480)
    p = ob.params_from_config(640, 480, vectors_needed=1)
    want = ob.scan_frames(p, mv, off, sd)
    want_nosd = ob.scan_frames(p, mv, off, None)
    for threads in (1, 2, 5, 37, 64):
        fl, sec = ob.bench_scan(p, mv, off, sd, nthreads=threads, reps=3)
        assert np.array_equal(fl, want) and sec > 0.0
        fl, _ = ob.bench_scan(p, mv, off, None, nthreads=threads, reps=1)
        assert np.array_equal(fl, want_nosd)
