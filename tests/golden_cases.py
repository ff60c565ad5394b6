"""Loaders for the committed golden fixtures (tests/golden/*.json), shared by the CPU tier
(oracle vs fixtures, tests/test_oracle.py) and the GPU tier (HIP path vs the SAME fixtures,
tests/test_gpu_golden.py)."""
import json
import os

import numpy as np

import mvtrim_amd as m

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build_mvs(case):
    rows = [list(r) for r in case.get("mvs", [])]
    for cx, cy, dx, dy, n in case.get("hits", []):
        x, y = 16 * cx + 8, 16 * cy + 8
        rows += [[x - dx, y - dy, x, y]] * n
    mv = np.zeros(len(rows), dtype=m.MV_DTYPE)
    if rows:
        a = np.array(rows, dtype=np.int64)
        mv["src_x"], mv["src_y"], mv["dst_x"], mv["dst_y"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    return mv


def load_hand_cases():
    g = json.load(open(os.path.join(GOLD, "check_frame_hand_cases.json")))
    out = []
    for c in g["cases"]:
        kw = dict(g["base"])
        kw.update(c.get("over", {}))
        out.append((c["name"], kw, c))
    return g, out


def load_merge_cases():
    g = json.load(open(os.path.join(GOLD, "merge_hand_cases.json")))
    return [(c["name"], dict(g["base"], **c.get("over", {})), c) for c in g["cases"]]


def merge_case_ts(case):
    ts = case.get("ts")
    if ts is None:
        a, b, s = case["ts_range"]
        ts = list(np.arange(a, b, s, dtype=np.float64))
    return [float(t) for t in ts]


def load_survey_segments():
    g = json.load(open(os.path.join(GOLD, "survey_segments.json")))
    ts = []
    for a, b in g["motion_frame_runs"]:
        ts += [float(3000 * i) * (1.0 / g["tb_den"]) for i in range(a, b + 1)]
    mp = m.MergeParams(duration=g["duration"], max_gap_sec=g["max_gap_sec"], padding_sec=g["padding_sec"],
                       min_savings_pct=5.0)
    return g, ts, mp


def load_filter_cases():
    return json.load(open(os.path.join(GOLD, "frame_filter_hand_cases.json")))


def filter_case_ticks(g, case):
    """AVFrame::pts of the frames the decoder hands over after the seek: frames first..first+n-1."""
    return [g["ticks_per_frame"] * (case["first"] + i) for i in range(case["n_frames"])]


def id_of(x):
    return x if isinstance(x, str) else None
