#!/usr/bin/env python3
"""Generates tests/golden/motion_scalar_golden.json by running the REFERENCE tool
(oracle/_ref/motion_scalar, built from /root/reference/tools/motion_scalar.cpp by
`make -C oracle ref`) on a JSON written in the extract_mvs schema from a small
deterministic synthetic stream.  Only runs where the reference tree is mounted; the
committed output is a data fixture (inputs are regenerated from the spec below)."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mvtrim_amd as m  # noqa: E402
from mvtrim_amd import synth  # noqa: E402

SPEC = dict(width=160, height=96, block=16, sub=2, fps=25.0, tb_den=90000, gop=12, seed=42,
            salt_p=0.02, oob_p=0.01)
N_FRAMES = 80
EVENTS = [[5, 30, 2, 1, 3, 2, 9, -3], [50, 70, 5, 3, 2, 2, -6, 5]]


def build():
    spec = synth.StreamSpec(**SPEC)
    spec.events = [synth.Event(*e) for e in EVENTS]
    frames = [synth.gen_frame(spec, i) for i in range(N_FRAMES)]
    pts = [spec.pts_seconds(i) for i in range(N_FRAMES)]
    return spec, frames, pts


def main():
    ref = os.path.join(ROOT, "oracle", "_ref", "motion_scalar")
    if not os.path.exists(ref):
        raise SystemExit("oracle/_ref/motion_scalar missing: run `make -C oracle ref` where /root/reference exists")
    spec, frames, pts = build()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "mv.json")
        m.mvjson.write_json(path, frames, pts, (1, spec.tb_den))
        out = subprocess.run([ref, path], check=True, capture_output=True, text=True).stdout
    rows = sorted((int(a), b) for a, b in (ln.split(",") for ln in out.strip().splitlines()[1:]))
    gold = {"_about": "stdout of the reference's tools/motion_scalar.cpp (rows sorted by second; the tool prints "
                      "them in unordered_map order) for the stream that make_motion_scalar_golden.py regenerates",
            "spec": SPEC, "n_frames": N_FRAMES, "events": EVENTS, "csv_header": out.splitlines()[0],
            "rows": [[a, b] for a, b in rows]}
    json.dump(gold, open(os.path.join(ROOT, "tests", "golden", "motion_scalar_golden.json"), "w"), indent=1)
    print(gold["rows"])


if __name__ == "__main__":
    main()
