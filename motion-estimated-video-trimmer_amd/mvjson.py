"""Reader / writer for the JSON that the reference's tools/extract_mvs.cpp prints
(schema at extract_mvs.cpp:97-169) — the reference's only on-disk MV format and the
input of tools/motion_scalar.cpp (BASELINE.json config 0).

    {"input": "...", "time_base": "num/den", "frames": [
        {"frame_index": 1-based, "pts_seconds": %.6f | null, "frame_type": "I|P|B|?",
         "num_mvs": n, "motion_vectors": [
            {"dst_x","dst_y","src_x"(%.3f),"src_y"(%.3f),"w","h","motion_x","motion_y",
             "motion_scale","source"}, ...]}, ...]}

The JSON does NOT carry AVMotionVector::src_x/src_y (it prints dst + motion/scale as a
float instead) nor `flags`.  read_json therefore rebuilds the int16 src fields as
dst + trunc(motion / scale) — C integer division, which is how upstream FFmpeg fills them
(libavcodec add_mb; third-party behaviour recalled, not verifiable offline) — and callers
that need byte-exact records should use mvfile.py instead.
"""
import json

import numpy as np

from ._abi import MV_DTYPE


def _fmt3(x):
    return "%.3f" % x


def write_json(path, frames, pts_seconds, time_base=(1, 90000), input_name="synthetic", frame_types=None):
    """frames[i]: MV_DTYPE array or None; pts_seconds[i]: float or None (-> null)."""
    with open(path, "w") as out:
        out.write("{\n")
        out.write('  "input": %s,\n' % json.dumps(input_name))
        out.write('  "time_base": "%d/%d",\n' % time_base)
        out.write('  "frames": [\n')
        for i, f in enumerate(frames):
            if i:
                out.write(",\n")
            n = 0 if f is None else len(f)
            out.write("    {\n")
            out.write('      "frame_index": %d,\n' % (i + 1))
            p = pts_seconds[i]
            out.write('      "pts_seconds": %s,\n' % ("null" if p is None or p < 0 else "%.6f" % p))
            ft = frame_types[i] if frame_types else ("I" if f is None else "P")
            out.write('      "frame_type": "%s",\n' % ft)
            out.write('      "num_mvs": %d,\n' % n)
            out.write('      "motion_vectors": [')
            for j in range(n):
                m = f[j]
                sc = int(m["motion_scale"]) or 1
                sx = int(m["dst_x"]) + float(int(m["motion_x"])) / sc       # extract_mvs.cpp:142-145
                sy = int(m["dst_y"]) + float(int(m["motion_y"])) / sc
                out.write("%s\n        {" % ("," if j else ""))
                out.write('"dst_x":%d,"dst_y":%d,"src_x":%s,"src_y":%s,"w":%d,"h":%d,'
                          '"motion_x":%d,"motion_y":%d,"motion_scale":%d,"source":%d}' %
                          (m["dst_x"], m["dst_y"], _fmt3(sx), _fmt3(sy), m["w"], m["h"], m["motion_x"],
                           m["motion_y"], m["motion_scale"], m["source"]))
            if n:
                out.write("\n      ")
            out.write("]\n    }")
        out.write("\n  ]\n}\n")


def read_json(path):
    """Returns (frames list of MV_DTYPE arrays / None for frames with zero MVs,
    pts_seconds list (None for null), time_base (num, den))."""
    root = json.load(open(path))
    num, den = (int(x) for x in root.get("time_base", "1/1").split("/"))
    frames, pts = [], []
    for fr in root["frames"]:
        mvs = fr.get("motion_vectors", [])
        pts.append(fr.get("pts_seconds"))
        if not mvs:
            frames.append(None)           # extract_mvs prints num_mvs 0 for frames without side data
            continue
        a = np.zeros(len(mvs), dtype=MV_DTYPE)
        for j, m in enumerate(mvs):
            sc = int(m["motion_scale"])
            a[j]["dst_x"], a[j]["dst_y"] = m["dst_x"], m["dst_y"]
            a[j]["w"], a[j]["h"] = m["w"], m["h"]
            a[j]["motion_x"], a[j]["motion_y"], a[j]["motion_scale"] = m["motion_x"], m["motion_y"], sc
            a[j]["source"] = m["source"]
            d = sc if sc else 1
            a[j]["src_x"] = m["dst_x"] + int(m["motion_x"] / d)            # trunc toward zero, like C
            a[j]["src_y"] = m["dst_y"] + int(m["motion_y"] / d)
        frames.append(a)
    return frames, pts, (num, den)
