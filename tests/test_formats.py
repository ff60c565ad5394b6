"""CPU tests of the MV file formats (mvfile.py .mtmv container, mvjson.py extract_mvs JSON)
and of the BASELINE.json config-0 plumbing: synthetic MV frames -> JSON in the reference's
schema -> the reference's own motion_scalar tool (golden output committed; re-run live when
oracle/_ref/motion_scalar exists) vs the oracle's restatement."""
import importlib.util
import json
import os
import subprocess

import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import synth

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load_maker():
    spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_motion_scalar_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def scalar_rows(frames, pts):
    b = m.FrameBatch.from_frames(frames)
    pts6 = [float("%.6f" % p) for p in pts]                     # the JSON carries %.6f seconds
    n_sec = int(max(pts6)) + 1
    acc = ob.motion_scalar(b.mv, b.frame_off, pts6, n_sec)
    touched = sorted({int(np.floor(p)) for p in pts6})          # the tool creates an entry per visited second
    return [[s, "%g" % acc[s]] for s in touched]


def test_motion_scalar_matches_reference_golden():
    gold = json.load(open(os.path.join(GOLD, "motion_scalar_golden.json")))
    mk = load_maker()
    assert gold["spec"] == mk.SPEC and gold["n_frames"] == mk.N_FRAMES
    spec, frames, pts = mk.build()
    got = scalar_rows(frames, pts)
    # seconds whose frames carry no MV at all still exist in the oracle's list; the tool only
    # creates a map entry when it adds a term
    want = [[int(a), b] for a, b in gold["rows"]]
    got = [r for r in got if r[1] != "0" or r in want]
    assert got == want


def test_motion_scalar_live_reference_tool(tmp_path):
    ref = os.path.join(ROOT, "oracle", "_ref", "motion_scalar")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/motion_scalar not built (needs the reference tree)")
    spec = synth.StreamSpec(width=320, height=240, block=16, sub=1, fps=30.0, gop=10, seed=9, salt_p=0.05)
    spec.events = [synth.Event(3, 40, 4, 3, 5, 4, 11, -7)]
    frames = [synth.gen_frame(spec, i) for i in range(70)]
    pts = [spec.pts_seconds(i) for i in range(70)]
    path = str(tmp_path / "mv.json")
    m.mvjson.write_json(path, frames, pts, (1, spec.tb_den))
    out = subprocess.run([ref, path], check=True, capture_output=True, text=True).stdout
    rows = sorted([int(a), b] for a, b in (ln.split(",") for ln in out.strip().splitlines()[1:]))
    assert out.splitlines()[0] == "second,motion_value"
    assert rows == [r for r in scalar_rows(frames, pts) if r[1] != "0"]


def test_json_roundtrip(tmp_path):
    spec = synth.StreamSpec(width=160, height=96, block=16, sub=2, fps=25.0, gop=6, seed=3)
    spec.events = [synth.Event(1, 9, 2, 1, 3, 2, 8, 3)]
    frames = [synth.gen_frame(spec, i) for i in range(10)]
    pts = [spec.pts_seconds(i) for i in range(10)]
    path = str(tmp_path / "mv.json")
    m.mvjson.write_json(path, frames, pts, (1, 90000))
    root = json.load(open(path))                                 # valid JSON in the reference schema
    assert root["time_base"] == "1/90000" and len(root["frames"]) == 10
    assert root["frames"][1]["frame_index"] == 2 and root["frames"][0]["num_mvs"] == 0
    assert set(root["frames"][1]["motion_vectors"][0]) == {"dst_x", "dst_y", "src_x", "src_y", "w", "h",
                                                           "motion_x", "motion_y", "motion_scale", "source"}
    f2, p2, tb = m.mvjson.read_json(path)
    assert tb == (1, 90000)
    for a, b in zip(frames, f2):
        if a is None:
            assert b is None
            continue
        for k in ("dst_x", "dst_y", "src_x", "src_y", "w", "h", "motion_x", "motion_y", "motion_scale", "source"):
            assert np.array_equal(a[k], b[k]), k                # src rebuilt as dst + trunc(motion / scale)
    assert [float("%.6f" % x) for x in pts] == p2


def test_mtmv_roundtrip(tmp_path):
    spec = synth.StreamSpec(width=320, height=240, block=16, sub=2, fps=30.0, gop=5, seed=8)
    frames = [synth.gen_frame(spec, i) for i in range(12)]
    frames[3] = np.zeros(0, dtype=m.MV_DTYPE)                     # side data with zero records
    ticks = [spec.pts_ticks(i) for i in range(12)]
    path = str(tmp_path / "s.mtmv")
    m.mvfile.write_mtmv(path, 320, 240, 1, 90000, 30.0, 0.4, ticks, frames)
    hdr, tab, mv = m.mvfile.read_mtmv(path)
    assert (int(hdr["width"]), int(hdr["n_frames"]), float(hdr["duration"])) == (320, 12, 0.4)
    assert tab["pts"].tolist() == ticks and tab["has_sd"].tolist() == [0 if f is None else 1 for f in frames]
    assert tab["key"].tolist() == [1 if f is None else 0 for f in frames]
    back = m.mvfile.frames_of(tab, mv)
    for a, b in zip(frames, back):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.tobytes() == np.asarray(b).tobytes()
    assert os.path.getsize(path) == 56 + 24 * 12 + 40 * int(hdr["n_records"])


def test_formats_feed_the_scan_identically(tmp_path):
    """Frames that went through the .mtmv container, and through the extract_mvs JSON (whose
    int16 src fields are rebuilt as dst + trunc(motion/scale)), give the oracle the same flags as
    the original records."""
    spec = synth.StreamSpec(width=640, height=480, block=16, sub=2, fps=25.0, gop=10, seed=31)
    spec.events = [synth.Event(2, 14, 10, 8, 4, 3, 9, -4), synth.Event(20, 28, 25, 15, 3, 3, -7, 2)]
    n = 30
    frames = [synth.gen_frame(spec, i) for i in range(n)]
    pts = [spec.pts_seconds(i) for i in range(n)]
    p = ob.params_from_config(640, 480)
    b0 = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b0.mv, b0.frame_off, b0.has_sd)
    assert 0 < want.sum() < n
    mt = str(tmp_path / "s.mtmv")
    m.mvfile.write_mtmv(mt, 640, 480, 1, 90000, 25.0, n / 25.0, [spec.pts_ticks(i) for i in range(n)], frames)
    _, tab, mv = m.mvfile.read_mtmv(mt)
    b1 = m.FrameBatch.from_frames(m.mvfile.frames_of(tab, mv))
    assert np.array_equal(ob.scan_frames(p, b1.mv, b1.frame_off, b1.has_sd), want)
    js = str(tmp_path / "s.json")
    m.mvjson.write_json(js, frames, pts, (1, 90000))
    f2, _, _ = m.mvjson.read_json(js)
    b2 = m.FrameBatch.from_frames(f2)
    assert np.array_equal(ob.scan_frames(p, b2.mv, b2.frame_off, b2.has_sd), want)
