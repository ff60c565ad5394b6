"""`.mtmv` — a compact binary container for extracted motion vectors, so the scan path
can run end to end on machines without FFmpeg.

The reference's only on-disk MV format is the JSON that tools/extract_mvs.cpp prints
(see mvjson.py); this is the same information, byte-exact and mmap-able:

    header   56 B  magic "MTMV1\\0\\0\\0", u32 width, height, tb_num, tb_den, f64 fps, f64 duration,
                   u64 n_frames, u64 n_records
    frames   24 B each  i64 pts (time_base units, AVFrame::pts), u64 rec_off (records),
                   u32 n_rec, u8 has_sd (AV_FRAME_DATA_MOTION_VECTORS present), u8 key (keyframe:
                   a backward seek may land here), u8 pad[2]
    records  40 B each  AVMotionVector bytes exactly as FFmpeg exports them

Little-endian.  The C++ reader is MtmvFile / MtmvSource in csrc/host/mtgpu_host.hpp.
"""
import numpy as np

from ._abi import MV_DTYPE

MAGIC = b"MTMV1\0\0\0"
HEADER_DTYPE = np.dtype([("magic", "S8"), ("width", "<u4"), ("height", "<u4"), ("tb_num", "<u4"),
                         ("tb_den", "<u4"), ("fps", "<f8"), ("duration", "<f8"), ("n_frames", "<u8"),
                         ("n_records", "<u8")])
FRAME_DTYPE = np.dtype([("pts", "<i8"), ("rec_off", "<u8"), ("n_rec", "<u4"), ("has_sd", "u1"),
                        ("key", "u1"), ("pad", "u1", (2,))])
assert HEADER_DTYPE.itemsize == 56 and FRAME_DTYPE.itemsize == 24


def write_mtmv(path, width, height, tb_num, tb_den, fps, duration, pts_ticks, frames, key=None):
    """frames[i]: MV_DTYPE array, or None when frame i has no MV side data.  key[i]: keyframe
    flags (default: frames without side data are keyframes, as I-frames export no MVs)."""
    n = len(frames)
    tab = np.zeros(n, dtype=FRAME_DTYPE)
    off = 0
    for i, f in enumerate(frames):
        k = 0 if f is None else len(f)
        tab[i] = (int(pts_ticks[i]), off, k, 0 if f is None else 1,
                  (1 if f is None else 0) if key is None else int(key[i]), (0, 0))
        off += k
    hdr = np.zeros(1, dtype=HEADER_DTYPE)
    hdr[0] = (MAGIC, width, height, tb_num, tb_den, fps, duration, n, off)
    with open(path, "wb") as fh:
        fh.write(hdr.tobytes())
        fh.write(tab.tobytes())
        for f in frames:
            if f is not None and len(f):
                fh.write(np.ascontiguousarray(f, dtype=MV_DTYPE).tobytes())


def read_mtmv(path):
    """Returns (header record, frame table, records memmap as MV_DTYPE)."""
    hdr = np.fromfile(path, dtype=HEADER_DTYPE, count=1)[0]
    if bytes(hdr["magic"]).ljust(8, b"\0") != MAGIC:
        raise ValueError(f"{path}: not an mtmv file")
    n = int(hdr["n_frames"])
    tab = np.fromfile(path, dtype=FRAME_DTYPE, count=n, offset=HEADER_DTYPE.itemsize)
    rec_off = HEADER_DTYPE.itemsize + FRAME_DTYPE.itemsize * n
    nrec = int(hdr["n_records"])
    mv = np.memmap(path, dtype=MV_DTYPE, mode="r", offset=rec_off, shape=(nrec,)) if nrec else \
        np.zeros(0, dtype=MV_DTYPE)
    return hdr, tab, mv


def frames_of(tab, mv):
    """Per-frame list (MV_DTYPE views or None) from a frame table."""
    out = []
    for r in tab:
        if not r["has_sd"]:
            out.append(None)
        else:
            a = int(r["rec_off"])
            out.append(mv[a:a + int(r["n_rec"])])
    return out
