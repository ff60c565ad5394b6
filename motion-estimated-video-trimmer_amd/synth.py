"""Deterministic synthetic AVMotionVector streams.

There is no decoder and no sample media in this environment, so the bench and the
parity tests run on synthetic MV arrays shaped like what FFmpeg's `export_mvs`
produces for CCTV footage: one record per partition in macroblock raster order,
`dst` = block centre, near-zero background motion, a few moving rectangles
("events") inside scripted time windows, isolated salt noise that must not
trigger, a few records pointing outside the frame, and periodic I-frames without
side data.  Only the *shape* follows FFmpeg's producer conventions; parity is never
derived from them.

All randomness is a counter-based splitmix64 hash of (seed, frame, record), so any
frame can be regenerated independently and bit-identically on any machine.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from ._abi import MV_DTYPE

_U64 = np.uint64


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finalizer on uint64 arrays (wrapping arithmetic)."""
    x = (x + _U64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> _U64(27))) * _U64(0x94D049BB133111EB)
    return x ^ (x >> _U64(31))


def _hash(seed: int, frame: int, idx: np.ndarray, salt: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _mix(np.array([(seed * 0x100000001B3 + salt) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
                    ^ _U64((frame * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF))
        return _mix(idx.astype(np.uint64) ^ base)


@dataclass
class Event:
    """A rectangle of cells moving with MV (dx,dy) during frames [f0, f1)."""
    f0: int
    f1: int
    cx: int
    cy: int
    cw: int
    ch: int
    dx: int
    dy: int


@dataclass
class StreamSpec:
    width: int = 1920
    height: int = 1080
    block: int = 16            # macroblock size of the producer (grid cell of the generator)
    sub: int = 2               # sub x sub partitions per block: 2 -> "dense8x8", 1 -> "dense16"
    fps: float = 30.0
    tb_den: int = 90000        # time_base = 1/tb_den
    gop: int = 30              # every gop-th frame is an I-frame without MV side data
    seed: int = 1
    salt_p: float = 1e-3       # isolated noisy cells per cell per frame
    oob_p: float = 1e-3        # records whose dst lies outside the frame
    events: List[Event] = field(default_factory=list)
    # records per cell INSIDE event rectangles and salt cells (sub == 1 only): a moving region of real footage is
    # coded with more partitions and two prediction directions, so a fine-grid cell there carries several MVs.
    # 4 = two directions (source -1 / +1) x two partitions: the only way a one-record-per-4x4-block stream can
    # reach VECTORS_NEEDED = 4 of the shipped env (config/motion_trim.env:75).  Frames then differ in size.
    event_records: int = 1

    @property
    def cells_x(self) -> int:
        return (self.width + self.block - 1) // self.block

    @property
    def cells_y(self) -> int:
        return (self.height + self.block - 1) // self.block

    @property
    def records_per_frame(self) -> int:
        return self.cells_x * self.cells_y * self.sub * self.sub

    def pts_ticks(self, f: int) -> int:
        return int(round(f * self.tb_den / self.fps))

    def pts_seconds(self, f: int) -> float:
        # exactly the reference's expression: frame->pts * av_q2d(time_base)  (motion_scanner.cpp:361)
        return float(self.pts_ticks(f)) * (1.0 / float(self.tb_den))


def spec_1080p(seed=1, sub=2, **kw) -> StreamSpec:
    return StreamSpec(1920, 1080, 16, sub, seed=seed, **kw)


def spec_4k(seed=1, sub=2, **kw) -> StreamSpec:
    return StreamSpec(3840, 2160, 16, sub, seed=seed, **kw)


def spec_4k_fine(seed=1, **kw) -> StreamSpec:
    """HEVC-style 4x4 MV granularity: one record per 4x4 block (960x540 grid)."""
    return StreamSpec(3840, 2160, 4, 1, seed=seed, **kw)


def spec_4k_fine_dense(seed=1, **kw) -> StreamSpec:
    """The same grid with 4 records per 4x4 block inside moving regions (2 directions x 2 partitions):
    the density at which the shipped env's VECTORS_NEEDED = 4 can say yes AND no on this grid.
    Salt cells carry 4 records too (an ACTIVE cell without an active neighbour must not trigger); their rate
    is lowered to ~10 per frame so that two of them are practically never adjacent on 518 400 cells."""
    kw.setdefault("salt_p", 2e-5)
    return StreamSpec(3840, 2160, 4, 1, seed=seed, event_records=4, **kw)


def scripted_events(spec: StreamSpec, n_frames: int, seed: Optional[int] = None) -> List[Event]:
    """Event windows separated by gaps on both sides of MAX_GAP_SEC (5 s): active for
    ~2 s, then alternately ~3 s and ~8 s of stillness."""
    rng = np.random.RandomState((spec.seed if seed is None else seed) & 0x7FFFFFFF)
    ev, f, long_gap = [], int(1.0 * spec.fps), False
    while f < n_frames:
        length = int(rng.randint(int(0.5 * spec.fps) + 1, int(2.5 * spec.fps) + 2))
        cw, ch = int(rng.randint(2, 7)), int(rng.randint(2, 6))
        cx = int(rng.randint(1, max(2, spec.cells_x - cw - 1)))
        cy = int(rng.randint(max(1, spec.cells_y // 10), max(2, spec.cells_y - ch - spec.cells_y // 10)))
        dx = int(rng.choice([-1, 1]) * rng.randint(4, 13))
        dy = int(rng.randint(-6, 7))
        ev.append(Event(f, min(n_frames, f + length), cx, cy, cw, ch, dx, dy))
        f += length + int((8.0 if long_gap else 3.0) * spec.fps)
        long_gap = not long_gap
    return ev


def gen_frame(spec: StreamSpec, f: int) -> Optional[np.ndarray]:
    """Records of frame f (MV_DTYPE), or None for an I-frame (no side data)."""
    if spec.gop > 0 and f % spec.gop == 0:
        return None
    cxn, cyn, sub, blk = spec.cells_x, spec.cells_y, spec.sub, spec.block
    n = spec.records_per_frame
    r = np.arange(n, dtype=np.int64)
    q = r % (sub * sub)
    cell = r // (sub * sub)
    mx, my = cell % cxn, cell // cxn
    part = blk // sub
    dst_x = mx * blk + part // 2 + part * (q % sub)
    dst_y = my * blk + part // 2 + part * (q // sub)

    h1 = _hash(spec.seed, f, r, 1)
    dx = (h1 % _U64(3)).astype(np.int64) - 1            # background: U{-1,0,1}
    dy = ((h1 >> _U64(8)) % _U64(3)).astype(np.int64) - 1

    # salt: isolated cells whose every partition carries a large vector
    hc = _hash(spec.seed, f, cell, 2)
    salt = (hc >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) < spec.salt_p
    sdx = ((hc >> _U64(3)) % _U64(17)).astype(np.int64) + 5
    dx = np.where(salt, sdx, dx)
    dy = np.where(salt, -sdx // 2, dy)

    for e in spec.events:
        if e.f0 <= f < e.f1:
            drift = ((f - e.f0) * e.dx) // (4 * blk)     # the rectangle drifts with its motion
            inside = (mx >= e.cx + drift) & (mx < e.cx + drift + e.cw) & (my >= e.cy) & (my < e.cy + e.ch)
            dx = np.where(inside, e.dx, dx)
            dy = np.where(inside, e.dy, dy)

    # a few records pointing outside the frame (padding macroblocks): bounds path
    h3 = _hash(spec.seed, f, r, 3)
    oob = (h3 >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) < spec.oob_p
    kind = (h3 % _U64(4)).astype(np.int64)
    dst_x = np.where(oob & (kind == 0), -1 - (r % 37), dst_x)
    dst_x = np.where(oob & (kind == 1), spec.width + blk * 2 + (r % 29), dst_x)
    dst_y = np.where(oob & (kind == 2), -3 - (r % 31), dst_y)
    dst_y = np.where(oob & (kind == 3), spec.height + blk * 2 + (r % 23), dst_y)
    big = np.where(oob, 9, 0)                            # and they do move (mag above threshold)
    dx = np.where(oob, big, dx)

    source = np.full(n, -1, dtype=np.int64)
    if spec.event_records > 1 and sub == 1:
        # cells of moving regions (events, salt) carry event_records records, in place (raster order kept):
        # record j of such a cell has direction -1 / +1 alternating and sits in partition j // 2 of the cell
        busy = salt.copy()
        for e in spec.events:
            if e.f0 <= f < e.f1:
                drift = ((f - e.f0) * e.dx) // (4 * blk)
                busy |= (mx >= e.cx + drift) & (mx < e.cx + drift + e.cw) & (my >= e.cy) & (my < e.cy + e.ch)
        busy &= ~oob
        counts = np.where(busy, spec.event_records, 1)
        idx = np.repeat(r, counts)
        j = np.arange(len(idx), dtype=np.int64) - np.repeat(np.cumsum(counts) - counts, counts)
        dst_x, dst_y, dx, dy, source = dst_x[idx], dst_y[idx], dx[idx], dy[idx], source[idx]
        multi = busy[idx]
        source = np.where(multi & (j % 2 == 1), 1, source)
        dst_x = np.where(multi, mx[idx] * blk + ((j // 2) * 2 + 1) % blk, dst_x)
        n = len(idx)

    out = np.zeros(n, dtype=MV_DTYPE)
    out["source"] = source.astype(np.int32)
    out["w"] = part
    out["h"] = part
    out["dst_x"] = dst_x.astype(np.int16)
    out["dst_y"] = dst_y.astype(np.int16)
    out["src_x"] = (dst_x - dx).astype(np.int16)
    out["src_y"] = (dst_y - dy).astype(np.int16)
    out["flags"] = 0
    out["motion_scale"] = 4
    out["motion_x"] = (-dx * 4).astype(np.int32)         # src = dst + motion/scale (extract_mvs.cpp:142-145)
    out["motion_y"] = (-dy * 4).astype(np.int32)
    return out


def gen_stream(spec: StreamSpec, n_frames: int, first_frame: int = 0):
    """Frames [first_frame, first_frame + n_frames) as CSR arrays:
    (mv MV_DTYPE[N], frame_off uint64[F+1], pts float64[F], has_sd uint8[F])."""
    frames = [gen_frame(spec, first_frame + i) for i in range(n_frames)]
    counts = np.array([0 if x is None else len(x) for x in frames], dtype=np.uint64)
    off = np.zeros(n_frames + 1, dtype=np.uint64)
    np.cumsum(counts, out=off[1:])
    mv = np.zeros(int(off[-1]), dtype=MV_DTYPE)
    for i, x in enumerate(frames):
        if x is not None:
            mv[int(off[i]):int(off[i + 1])] = x
    pts = np.array([spec.pts_seconds(first_frame + i) for i in range(n_frames)], dtype=np.float64)
    has_sd = np.array([0 if x is None else 1 for x in frames], dtype=np.uint8)
    return mv, off, pts, has_sd


def random_frames(rng: np.random.RandomState, n_frames: int, max_records: int, width: int,
                  height: int, hot: float = 0.3) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Ragged adversarial frames for edge-case parity: arbitrary record order, heavy cell
    collisions, empty frames, frames without side data, extreme coordinates.
    |dx|,|dy| stay <= 32767 (the reference's defined domain)."""
    counts = rng.randint(0, max_records + 1, size=n_frames)
    counts[rng.rand(n_frames) < 0.15] = 0
    off = np.zeros(n_frames + 1, dtype=np.uint64)
    np.cumsum(counts, out=off[1:])
    n = int(off[-1])
    mv = np.zeros(n, dtype=MV_DTYPE)
    has_sd = (rng.rand(n_frames) < 0.9).astype(np.uint8)
    if n:
        style = rng.rand(n)
        dst_x = rng.randint(-64, width + 64, size=n)
        dst_y = rng.randint(-64, height + 64, size=n)
        # hot spots: many vectors into few cells
        hx, hy = rng.randint(0, width, size=8), rng.randint(0, height, size=8)
        pick = rng.randint(0, 8, size=n)
        hotm = style < hot
        dst_x = np.where(hotm, hx[pick] + rng.randint(0, 40, size=n), dst_x)
        dst_y = np.where(hotm, hy[pick] + rng.randint(0, 40, size=n), dst_y)
        ext = style > 0.97
        dst_x = np.where(ext, rng.choice([-32768, -16384, 16383, 32767, 0], size=n), dst_x)
        dst_y = np.where(ext, rng.choice([-32768, -1, 0, 32767, 1], size=n), dst_y)
        dx = rng.randint(-6, 7, size=n)
        dy = rng.randint(-6, 7, size=n)
        far = rng.rand(n) < 0.02
        dx = np.where(far, rng.randint(-32767, 32768, size=n), dx)
        dy = np.where(far, rng.randint(-32767, 32768, size=n), dy)
        src_x = np.clip(dst_x - dx, -32768, 32767)
        src_y = np.clip(dst_y - dy, -32768, 32767)
        # keep |dst - src| <= 32767 after clipping
        src_x = np.where(np.abs(dst_x - src_x) > 32767, dst_x, src_x)
        src_y = np.where(np.abs(dst_y - src_y) > 32767, dst_y, src_y)
        mv["dst_x"], mv["dst_y"] = dst_x.astype(np.int16), dst_y.astype(np.int16)
        mv["src_x"], mv["src_y"] = src_x.astype(np.int16), src_y.astype(np.int16)
        mv["w"], mv["h"] = 8, 8
        mv["source"] = -1
        mv["motion_scale"] = 4
        # unrelated fields carry junk on purpose: the scan must ignore them
        mv["flags"] = rng.randint(0, 2 ** 62, size=n).astype(np.uint64)
        mv["motion_x"] = rng.randint(-2 ** 31, 2 ** 31 - 1, size=n)
        mv["motion_y"] = rng.randint(-2 ** 31, 2 ** 31 - 1, size=n)
    return mv, off, has_sd
