"""CPU tests of the host-side mirror (motion-estimated-video-trimmer_amd/scanner.py,
config.py, dist.py helpers) against the oracle.  No GPU needed."""
import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import dist as mdist
from mvtrim_amd import synth

import oracle_binding as ob


def test_params_from_config_matches_oracle():
    """libmtgpu's host-side derivation (mtgpu_params_from_config) == the oracle's, over many
    sizes / block settings / masks (reference src/motion_scanner.cpp:184-199)."""
    rng = np.random.RandomState(0)
    for _ in range(400):
        w, h = int(rng.randint(1, 8000)), int(rng.randint(1, 5000))
        sh = int(rng.randint(0, 7))
        bs = int(rng.choice([1 << sh, 16, 8, 4, 1]))
        vm = float(np.float32(rng.choice([0.0, 0.05, 0.1, 0.25, 0.5, 0.7, rng.rand()])))
        kw = dict(mv_threshold_sq=float(rng.choice([16.0, 4.0, 0.0, 7.5])), block_size=bs, block_shift=sh,
                  vectors_needed=int(rng.randint(-3, 600)), clusters_needed=int(rng.randint(-3, 50)),
                  vertical_mask=vm)
        try:
            want = ob.params_from_config(w, h, **kw)
        except ValueError:
            with pytest.raises(m.MtgpuError):
                m.ScanParams.from_config(w, h, **kw)
            continue
        assert m.ScanParams.from_config(w, h, **kw) == want


def test_config_getters_follow_env(monkeypatch):
    c = m.config
    for k in ["MV_THRESHOLD_SQ", "BLOCK_SIZE", "BLOCK_SHIFT", "VECTORS_NEEDED", "CLUSTERS_NEEDED",
              "VERTICAL_MASK", "MAX_GAP_SEC", "PADDING_SEC", "CHUNK_DURATION_SEC", "TARGET_FPS", "MIN_SAVINGS_PCT"]:
        monkeypatch.delenv(k, raising=False)
    c.forget()
    # code defaults (include/motion_trim/config.hpp:56-125)
    assert (c.mv_threshold_sq(), c.block_size(), c.block_shift(), c.vectors_needed(), c.clusters_needed()) == \
        (16.0, 16, 4, 2, 2)
    assert c.vertical_mask() == float(np.float32(0.05))
    assert (c.max_gap_sec(), c.padding_sec(), c.chunk_duration_sec(), c.target_fps(), c.min_savings_pct()) == \
        (5.0, 0.5, 30.0, 0.0, 5.0)
    p = m.ScanParams.from_config(1920, 1080)
    assert (p.mv_threshold_sq, p.vectors_needed, p.clusters_needed, p.vertical_margin) == (16.0, 2, 2, 3)
    # shipped env file values (config/motion_trim.env:36,75,93,113,211)
    monkeypatch.setenv("MV_THRESHOLD_SQ", "4.0")
    monkeypatch.setenv("VECTORS_NEEDED", "4")
    monkeypatch.setenv("TARGET_FPS", "10.0")
    # read once per process, like the reference's function-local statics (config.hpp:56-59): still the defaults
    assert (c.mv_threshold_sq(), c.vectors_needed(), c.target_fps()) == (16.0, 2, 0.0)
    c.forget()                                           # "a new process"
    p = m.ScanParams.from_config(1920, 1080)
    assert (p.mv_threshold_sq, p.vectors_needed) == (4.0, 4)
    assert c.target_fps() == 10.0
    monkeypatch.setenv("VECTORS_NEEDED", "260")          # uint8 cast
    c.forget()
    assert c.vectors_needed() == 4 and m.ScanParams.from_config(1920, 1080).vectors_needed == 4
    c.forget()
    mp = m.MergeParams(duration=60.0)
    assert (mp.max_gap_sec, mp.padding_sec, mp.min_savings_pct) == (5.0, 0.5, 5.0)


def test_frame_filter_and_chunks_match_oracle():
    rng = np.random.RandomState(3)
    for _ in range(200):
        fps = float(rng.choice([25.0, 30.0, 29.97, 50.0, 60.0]))
        tfps = float(rng.choice([0.0, 5.0, 10.0, 30.0, 100.0]))
        skip = m.frame_skip(fps, tfps)
        assert skip == ob.lib().mto_frame_skip(fps, tfps)
        tb = 1.0 / 90000.0
        n = int(rng.randint(0, 400))
        first = int(rng.randint(0, 1000))
        ticks = [int(round((first + i) * 90000 / fps)) for i in range(n)]
        start = float(rng.rand() * 20)
        end = start + float(rng.rand() * 10)
        assert m.filter_frames(ticks, tb, start, end, skip) == ob.filter_frames(ticks, tb, start, end, skip)
    for dur, ch in [(70.0, 30.0), (60.0, 30.0), (0.0, 30.0), (0.1, 60.0), (123.456, 7.5), (1e-9, 1.0)]:
        assert m.make_chunks(dur, ch) == ob.chunks(dur, ch)


def test_frame_batch_from_frames():
    a = np.zeros(3, dtype=m.MV_DTYPE)
    a["dst_x"] = [1, 2, 3]
    b = m.FrameBatch.from_frames([None, a, np.zeros(0, dtype=m.MV_DTYPE), a[:1]], pts=[0.0, 0.1, 0.2, 0.3])
    assert b.n_frames == 4 and b.frame_off.tolist() == [0, 0, 3, 3, 4]
    assert b.has_sd.tolist() == [0, 1, 1, 1] and b.mv["dst_x"].tolist() == [1, 2, 3, 1]
    assert m.MV_DTYPE.itemsize == 40 and b.mv.view(np.uint8).size == 160
    raw = a.view(np.uint8).reshape(3, 40)
    assert raw[1, 10] == 2 and raw[1, 11] == 0         # dst_x sits at byte 10 (AVMotionVector layout)


def test_synth_stream_is_deterministic_and_shaped():
    spec = synth.spec_1080p(seed=5)
    spec.events = synth.scripted_events(spec, 40)
    f1 = synth.gen_frame(spec, 7)
    f2 = synth.gen_frame(spec, 7)
    assert f1.tobytes() == f2.tobytes() and len(f1) == 32640 == spec.records_per_frame
    assert synth.gen_frame(spec, 30) is None           # I-frame
    assert spec.pts_seconds(301) == float(3000 * 301) * (1.0 / 90000.0)
    assert synth.spec_4k().records_per_frame == 129600 and synth.spec_4k_fine().records_per_frame == 518400
    assert np.abs(f1["dst_x"].astype(int) - f1["src_x"]).max() <= 32767


def test_sharding_helpers():
    for n, w in [(64, 8), (10, 3), (3, 8), (0, 4), (17, 1)]:
        rs = [mdist.shard_range(n, w, r) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in rs]
        assert max(sizes) - min(sizes) <= 1
    off = np.concatenate([[0], np.cumsum([100, 0, 100, 100, 0, 0, 100, 400, 100, 100])])
    cuts = mdist.shard_by_records(off, 4)
    assert cuts[0][0] == 0 and cuts[-1][1] == 10 and all(cuts[i][1] == cuts[i + 1][0] for i in range(3))
    recs = [int(off[b] - off[a]) for a, b in cuts]
    assert sum(recs) == 1000 and max(recs) <= 500


def test_launch_planner_known_answers_and_invariants(monkeypatch):
    """mtgpu_plan_preview: the launch planner (counter form, tile / band geometry, workgroup size)
    is host arithmetic — checked here without a GPU for the MI355X limits (160 KB LDS, 256 CUs)."""
    for k in ("MTGPU_FORCE_FB", "MTGPU_FORCE_BLOCK", "MTGPU_FORCE_CHUNK", "MTGPU_BAND_LDS_KB", "MTGPU_MAX_TILE_KB"):
        monkeypatch.delenv(k, raising=False)
    fine = dict(block_size=4, block_shift=2)
    cases = [
        ((1920, 1080, {}), dict(counter_bits=32, counter_mode=0, bands=1, block_threads=512, band_rows=62)),
        ((3840, 2160, {}), dict(counter_bits=32, counter_mode=0, bands=1, block_threads=1024, band_rows=123)),
        ((3840, 2160, dict(fine, vectors_needed=1)), dict(counter_bits=1, counter_mode=1, bands=1, band_rows=486)),
        ((3840, 2160, dict(fine, vectors_needed=2)), dict(counter_bits=2, counter_mode=1, bands=1, band_rows=486)),
        ((3840, 2160, dict(fine, vectors_needed=4)), dict(counter_bits=4, counter_mode=1, bands=2, band_rows=243)),
        ((3840, 2160, dict(fine, vectors_needed=9)), dict(counter_bits=8, counter_mode=2)),           # 8-bit CAS fields
        ((1920, 1080, dict(vectors_needed=200)), dict(counter_bits=32, bands=1)),                     # u32 counts: no packing needed
    ]
    for (w, h, kw), want in cases:
        plan = m.plan_preview(m.ScanParams.from_config(w, h, **kw))
        for key, v in want.items():
            assert plan[key] == v, (w, h, kw, key, plan)
    # the 1080p tile: 64 counter rows x 120 cells x 4 B + 64 mask rows x 2 words x 8 B + control words
    p1080 = m.plan_preview(m.ScanParams.from_config(1920, 1080))
    assert p1080["lds_bytes"] == 64 * 120 * 4 + 64 * 2 * 8 + 16 == 31760
    # invariants over random legal grids and every VECTORS_NEEDED: the automatic plan always exists
    # (MT_ERR_CAPACITY is unreachable), fits LDS, covers every analysed row, and a thermometer field
    # is never narrower than VECTORS_NEEDED
    rng = np.random.RandomState(7)
    for _ in range(400):
        sh = int(rng.randint(0, 7))
        w, h = int(rng.randint(1, 32767) << sh) if rng.rand() < 0.1 else int(rng.randint(8, 8000)), int(rng.randint(8, 5000))
        vn = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 64, 255]))
        try:
            p = m.ScanParams.from_config(w, h, block_size=1 << sh, block_shift=sh, vectors_needed=vn,
                                         vertical_mask=float(rng.choice([0.0, 0.05, 0.3])))
        except m.MtgpuError:
            continue                                          # grid beyond int16: rejected before planning
        plan = m.plan_preview(p)
        rows = max(1, p.grid_h - 2 * p.vertical_margin)
        assert 0 < plan["lds_bytes"] <= 163840, (w, h, sh, vn, plan)
        assert plan["bands"] * plan["band_rows"] >= rows and (plan["bands"] - 1) * plan["band_rows"] < rows
        assert 1 <= plan["chunk_rows"] <= plan["band_rows"] and plan["block_threads"] in (256, 512, 1024)
        if plan["counter_mode"] == 1:
            assert plan["counter_bits"] >= vn and plan["counter_bits"] in (1, 2, 4, 8)
        if plan["counter_mode"] == 2:
            assert plan["counter_bits"] == 8 and vn > 8
    # a smaller LDS (64 KB parts) only changes the geometry, never the validity
    small = m.plan_preview(m.ScanParams.from_config(3840, 2160), lds_bytes=65536, cu_count=104)
    assert small["lds_bytes"] <= 65536 and small["counter_bits"] in (1, 2, 4, 8, 32)
    with pytest.raises(m.MtgpuError):
        bad = m.ScanParams.from_config(1920, 1080)
        bad.grid_w = 0
        m.plan_preview(bad)


def test_replay_of_the_recorded_wrong_flag_configuration():
    """profiles/r04_soak_mismatch_with_registered_staging.txt (seed 10242, iteration 70, frame 33: the zero-copy pipe on
    hipHostRegister'ed staging returned 0 where the oracle says 1) replayed on the CPU from the soak's own random stream
    (tests/soak_replay.py; the soak fed the pipe every 5th iteration then).  What the record can and cannot mean is
    read off here instead of re-running the soak:
      - the reconstruction is the recorded configuration (949 x 146, shift 4, T 4.0, 8-bit CAS fields forced);
      - two batches through two staging blocks, no re-pin: frame 33 sat in the FIRST use of a freshly pinned block,
        so neither staging reuse, nor a recycled address, nor the grow path can be the cause;
      - frame 33 was position 1 of its batch: flag byte 1, written by the same lane that wrote byte 0 (two frames per
        workgroup) — and byte 0 (frame 32) arrived;
      - its answer is not marginal: 42 centre cells against a need of 2, no single lost vote flips it, a frame whose
        records all read as zeros would — the fault was wholesale (the result byte, or the frame's whole record range)."""
    import oracle_binding as ob
    from soak_replay import pipe_batches, replay
    head, tail, plan, p = replay(10242, 70, pipe_every=5)
    assert (head["w"], head["h"], head["kw"]["block_shift"], head["kw"]["mv_threshold_sq"]) == (949, 146, 4, 4.0)
    assert head["force_fb"] == 108 and plan["counter_mode"] == 2 and plan["bands"] == 1 and head["knobs"]["MTGPU_GROUP"] == "2"
    assert tail["pipe"] == (50000, 32, 2) and tail["n_frames"] == 64
    batches, grows = pipe_batches(tail["off"], tail["sd"], *tail["pipe"])
    assert grows == [] and [(slot, fr[0], fr[-1]) for slot, fr, _ in batches] == [(0, 0, 31), (1, 32, 63)]
    mv, off, sd = tail["mv"], tail["off"], tail["sd"]
    fr = mv[int(off[33]):int(off[34])]
    flag, centres, _ = ob.check_frame(p, fr, True, count_centres=True)
    assert (flag, centres) == (1, 42) and p.clusters_needed == 2
    assert ob.check_frame(p, mv[int(off[32]):int(off[33])], True) == 1
    assert all(ob.check_frame(p, np.delete(fr, i), True) == 1 for i in range(0, len(fr), 7))
    assert ob.check_frame(p, np.zeros_like(fr), True) == 0
