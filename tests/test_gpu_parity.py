"""GPU parity tests (run with `-m gpu` on the MI355X box): the HIP path, called
through the C ABI (libmtgpu.so), against the CPU oracle on identical inputs.

Bar: bit-exact.  Per-frame flags are integers; segment doubles are compared by
their 64-bit patterns.
"""
import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import synth

import oracle_binding as ob
from conftest import experiments_build

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def junk_padding(mv, rng):
    """Fill the padding bytes of the records (14-15, 34-39) with junk: the scan must ignore them."""
    raw = mv.view(np.uint8).reshape(-1, 40)
    if len(raw):
        raw[:, 14:16] = rng.randint(0, 256, size=(len(raw), 2))
        raw[:, 34:40] = rng.randint(0, 256, size=(len(raw), 6))
    return mv


def assert_scan_parity(scanner, params, mv, off, has_sd):
    want = ob.scan_frames(params, mv, off, has_sd)
    got = scanner.check_frames(m.FrameBatch(mv, off, None, has_sd))
    bad = np.flatnonzero(want != got)
    assert bad.size == 0, f"{bad.size} frames differ, first {bad[:8]}: want {want[bad[:8]]} got {got[bad[:8]]}"
    return want


# ------------------------------------------------------------------ scan: streams

@pytest.mark.parametrize("cfg", ["code_defaults", "shipped_env"])
@pytest.mark.parametrize("sub", [1, 2])
def test_scan_1080p_stream(gpu_scanner_factory, cfg, sub):
    kw = dict(m.config.CODE_DEFAULTS if cfg == "code_defaults" else m.config.SHIPPED_ENV)
    if sub == 1:
        kw["vectors_needed"] = 1               # one record per cell: a cell never collects 2 votes
    spec = synth.spec_1080p(seed=11 + sub, sub=sub)
    spec.events = synth.scripted_events(spec, 120)
    mv, off, pts, sd = synth.gen_stream(spec, 120)
    p = ob.params_from_config(1920, 1080, **kw)
    assert (p.grid_w, p.grid_h, p.vertical_margin) == (120, 68, 3)
    s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080, **kw))
    assert s.params == p
    want = assert_scan_parity(s, p, mv, off, sd)
    assert 0 < want.sum() < len(want)          # the stream has both motion and still frames
    # has_sd == NULL convention: side data iff >= 1 record (identical here)
    got2 = s.check_frames(m.FrameBatch(mv, off))
    assert np.array_equal(got2, want)


@pytest.mark.parametrize("force_fb", [None, 2, 32])
@pytest.mark.parametrize("cfg,sub", [("code_defaults", 2), ("shipped_env", 2), ("code_defaults", 1), ("shipped_env", 1)])
def test_scan_4k_stream(gpu_scanner_factory, force_fb, cfg, sub):
    """BASELINE config 3 (240x135) with BOTH parameter sets SURVEY 8(d) names — code defaults (T 16, VECTORS_NEEDED 2)
    and the shipped env (T 4, VECTORS_NEEDED 4: config/motion_trim.env:36, 75) — at four records per 16-px cell and at
    one (VECTORS_NEEDED 1 then: one record per cell never collects two votes), on the 124 KB tile of 32-bit counters
    the planner picks, on the forced packed form (2-bit fields where they can count VECTORS_NEEDED, the planner's
    narrowest packed form otherwise) and on forced 32-bit counters."""
    kw = dict(m.config.CODE_DEFAULTS if cfg == "code_defaults" else m.config.SHIPPED_ENV)
    if sub == 1:
        kw["vectors_needed"] = 1
    spec = synth.spec_4k(seed=5 + sub, sub=sub)
    spec.events = synth.scripted_events(spec, 16)
    spec.events.append(synth.Event(3, 9, 100, 60, 4, 3, 9, -5))
    mv, off, pts, sd = synth.gen_stream(spec, 16)
    p = ob.params_from_config(3840, 2160, **kw)
    assert (p.grid_w, p.grid_h, p.vertical_margin) == (240, 135, 6)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    if force_fb is None:
        assert s.plan["counter_bits"] == 32 and s.plan["bands"] == 1 and s.plan["lds_bytes"] > 120 * 1024
    elif force_fb == 2:
        assert s.plan["counter_bits"] == (2 if p.vectors_needed <= 2 else 4)
    want = assert_scan_parity(s, p, mv, off, sd)
    assert 6 <= want.sum() < len(want)


@pytest.mark.parametrize("force_fb", [None, 2, 8, 108, 32])
def test_scan_4k_fine_grid(gpu_scanner_factory, force_fb):
    """960x540 grid (BLOCK_SIZE=4, SHIFT=2).  Default plan (vectors_needed 1): 1-bit packed
    counters, whole grid in one LDS tile.  2: cluster test in row chunks.  8 / 108 (8-bit
    thermometer / CAS fields) -> row bands; 32: plain u32 counters -> many row bands."""
    spec = synth.spec_4k_fine(seed=9)
    spec.events = [synth.Event(1, 3, 300, 200, 6, 6, 9, 3),
                   synth.Event(2, 4, 500, 27, 5, 1, -7, 0),       # single row at y_min
                   synth.Event(1, 4, 100, 250, 3, 120, 9, 0)]     # tall: crosses chunk / band seams
    mv, off, pts, sd = synth.gen_stream(spec, 5)
    kw = dict(block_size=4, block_shift=2, vectors_needed=1)   # one record per 4x4 cell
    p = ob.params_from_config(3840, 2160, **kw)
    assert (p.grid_w, p.grid_h, p.vertical_margin) == (960, 540, 27)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    plan = s.plan
    if force_fb is None:
        assert plan["counter_bits"] == 1 and plan["bands"] == 1 and plan["counter_mode"] == 1
    elif force_fb == 2:
        assert plan["counter_bits"] == 2 and plan["bands"] == 1 and plan["chunk_rows"] < 486
    else:
        assert plan["counter_bits"] == force_fb % 100 and plan["bands"] > 1
    want = assert_scan_parity(s, p, mv, off, sd)
    assert want.sum() >= 3


@pytest.mark.parametrize("force_fb", [2, 4, 8, 108, 32])
def test_fine_grid_cluster_across_seams(gpu_scanner_factory, force_fb):
    """Two active cells stacked vertically exactly on a chunk seam / band seam: each side must
    see the other's row as a neighbour (halo rows)."""
    p = ob.params_from_config(3840, 2160, block_size=4, block_shift=2, vectors_needed=1, clusters_needed=2)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    plan = s.plan
    step = plan["band_rows"] if plan["bands"] > 1 else plan["chunk_rows"]
    assert step < 486
    seam = p.vertical_margin + step               # first centre row of the second band / chunk
    frames = []
    for (ya, yb) in [(seam - 1, seam), (seam, seam + 1), (seam - 2, seam - 1), (seam - 1, seam + 1)]:
        mv = np.zeros(2, dtype=m.MV_DTYPE)
        mv["dst_x"] = 400 * 4 + 2
        mv["dst_y"] = [ya * 4 + 2, yb * 4 + 2]
        mv["src_x"] = mv["dst_x"] - 9
        mv["src_y"] = mv["dst_y"]
        frames.append(mv)
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert list(want) == [1, 1, 1, 0]


def _fine_shipped_env_scanner(gpu_scanner_factory):
    kw = dict(m.config.SHIPPED_ENV)
    kw.update(block_size=4, block_shift=2)
    p = ob.params_from_config(3840, 2160, **kw)
    assert (p.grid_w, p.grid_h, p.vertical_margin, p.vectors_needed, p.clusters_needed) == (960, 540, 27, 4, 2)
    s = gpu_scanner_factory(p)
    plan = s.plan                                   # the AUTOMATIC plan: 4-bit thermometer fields, 2 spill bands
    assert (plan["counter_bits"], plan["counter_mode"], plan["bands"], plan["band_rows"]) == (4, 1, 2, 243)
    return p, s


def test_fine_grid_shipped_env_dense_stream_two_bands(gpu_scanner_factory):
    """BASELINE config 5 under the SHIPPED env (config/motion_trim.env:36,75: T=4, VECTORS_NEEDED=4) on input
    where that setting can say yes and no: 4 records per 4x4 block inside moving regions (2 directions x 2
    partitions; synth.spec_4k_fine_dense).  Events in band 0, in band 1 (every vote there arrives through the
    spill queue) and across the seam at rows 269 | 270; a region with only 3 records per block must stay
    inactive; full-size ragged frames; automatic 2-band plan."""
    p, s = _fine_shipped_env_scanner(gpu_scanner_factory)
    spec = synth.spec_4k_fine_dense(seed=11)
    spec.events = [synth.Event(1, 2, 300, 100, 5, 4, 9, 3),        # frame 1: band 0 only
                   synth.Event(2, 3, 640, 400, 4, 3, -7, 2),       # frame 2: band 1 only (replayed votes)
                   synth.Event(3, 4, 500, 268, 3, 4, 6, 0),        # frame 3: rows 268..271, across the seam
                   synth.Event(4, 5, 10, 269, 1, 2, 5, 0),         # frame 4: ONE column, rows 269 + 270: exactly 2 centres
                   synth.Event(5, 6, 700, 27, 6, 1, 8, 0),         # frame 5: a single row at y_min (horizontal pairs)
                   synth.Event(6, 7, 100, 512, 2, 1, 8, 0)]        # frame 6: last analysed row, band 1
    mv, off, pts, sd = synth.gen_stream(spec, 9)                   # frames 7, 8: no event (salt only) -> 0
    counts = np.diff(off.astype(np.int64))
    assert counts[0] == 0 and len(set(counts[1:].tolist())) > 3    # I-frame + ragged P-frames
    want = assert_scan_parity(s, p, mv, off, sd)
    assert want.tolist() == [0, 1, 1, 1, 1, 1, 1, 0, 0]
    # the same frames with ONE record of every busy block removed (3 votes < VECTORS_NEEDED): nothing fires
    keep = np.ones(len(mv), dtype=bool)
    cell = (mv["dst_y"].astype(np.int64) >> 2) * 960 + (mv["dst_x"].astype(np.int64) >> 2)
    for f in range(1, 9):
        a, b = int(off[f]), int(off[f + 1])
        c = cell[a:b]
        dup = np.flatnonzero((c[1:] == c[:-1]) & (mv["source"][a + 1:b] == 1) & (mv["source"][a:b - 1] == -1)) + 1
        first_of_four = dup[::2]                                    # records 1 and 3 of a block are "+1": drop record 1
        keep[a + first_of_four] = False
    mv3 = mv[keep]
    off3 = np.concatenate([[0], np.cumsum([keep[int(off[f]):int(off[f + 1])].sum() for f in range(9)])]).astype(np.uint64)
    assert (off[-1] - off3[-1]) >= 60
    want3 = assert_scan_parity(s, p, mv3, off3, sd)
    assert want3.sum() == 0


def test_band_seam_cluster_needs_every_replayed_vote(gpu_scanner_factory):
    """Hand-built frames on the automatic 2-band plan (centres [27,270) | [270,513), queue from row 269 on):
    a vertical pair (x,269)+(x,270) has exactly CLUSTERS_NEEDED = 2 centres, one counted by each band, and
    every cell holds exactly VECTORS_NEEDED = 4 votes — band 0 sees row 270 only as its halo, band 1 sees
    row 269 only through replayed queue entries.  One vote fewer anywhere, or one lost replay, flips the flag."""
    p, s = _fine_shipped_env_scanner(gpu_scanner_factory)

    def frame(cells, filler=20000):
        """cells: [(gx, gy, votes)]; `filler` still records around them so that the votes sit deep inside a
        large frame (many streaming steps, many waves appending to the queue)."""
        rng = np.random.RandomState(len(cells) * 7 + cells[0][1])
        recs = []
        for gx, gy, votes in cells:
            recs += [(gx * 4 + 1 + (i & 1) * 2, gy * 4 + 2, 3) for i in range(votes)]
        n = filler + len(recs)
        mv = np.zeros(n, dtype=m.MV_DTYPE)
        mv["dst_x"] = rng.randint(0, 3840, size=n)
        mv["dst_y"] = rng.randint(0, 2160, size=n)
        mv["src_x"], mv["src_y"] = mv["dst_x"] - 1, mv["dst_y"]            # |d|^2 = 1 < 4: below the threshold
        where = rng.choice(n, size=len(recs), replace=False)
        for w, (x, y, d) in zip(where, recs):
            mv["dst_x"][w], mv["dst_y"][w] = x, y
            mv["src_x"][w], mv["src_y"][w] = x - d, y                      # |d|^2 = 9 >= 4
        return mv

    cases = [([(400, 269, 4), (400, 270, 4)], 1),        # the seam pair: 1 centre per band
             ([(400, 269, 4), (400, 270, 3)], 0),        # row 270 one vote short: no cell has an active neighbour
             ([(400, 269, 3), (400, 270, 4)], 0),        # row 269 one vote short (band 1's halo, replayed)
             ([(400, 270, 4), (400, 271, 4)], 1),        # both in band 1: only replayed votes
             ([(400, 270, 4), (400, 271, 3), (401, 270, 3)], 0),
             ([(400, 268, 4), (400, 269, 4)], 1),        # both in band 0; row 269 is queued as well, harmlessly
             ([(400, 269, 4), (401, 269, 4)], 1),        # horizontal pair on band 0's last centre row
             ([(400, 270, 4), (401, 270, 4)], 1),        # horizontal pair on band 1's first centre row
             ([(400, 269, 4), (401, 270, 4)], 0),        # diagonal across the seam: not 4-neighbours
             ([(400, 269, 9), (400, 270, 200)], 1),      # saturated fields (thermometer full) on both sides
             ([(0, 269, 4), (0, 270, 4), (959, 269, 4), (959, 270, 4)], 0),   # columns 0 / gw-1 are never centres
             ([(1, 269, 4), (0, 269, 4), (958, 270, 4), (959, 270, 4)], 1)]   # ... but they count as neighbours
    frames = [frame(c) for c, _ in cases]
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert want.tolist() == [w for _, w in cases]
    # and as 8-byte compact records through the pinned zero-copy pipe (what the host dispatcher stages)
    pipe = m.ScanPipe(s, 3 * 20100, 3, 2)
    for i, f in enumerate(frames):
        pipe.feed(f, float(i), tag=i)
    got8 = pipe.drain()
    pipe.close()
    assert [fl for _, fl, _ in got8] == want.tolist()


@pytest.mark.parametrize("force_fb", [1, 2, 4, 8, 108])
@pytest.mark.parametrize("vec", [1, 2, 3, 4, 5, 8, 9, 16, 255])
def test_packed_counter_forms(gpu_scanner_factory, force_fb, vec):
    """Every packed counter form on a 1080p grid (thermometer 1/2/4/8 bits, 8-bit CAS = 108),
    with heavy same-cell contention and cells sitting exactly at vectors_needed-1 /
    vectors_needed / far above."""
    if force_fb != 108 and vec > force_fb:
        pytest.skip("thermometer field too narrow for this vectors_needed (the planner never picks it)")
    p = ob.params_from_config(1920, 1080, vectors_needed=vec, clusters_needed=2)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    assert s.plan["counter_bits"] == force_fb % 100 and s.plan["counter_mode"] == (2 if force_fb == 108 else 1)
    rng = np.random.RandomState(vec * 10 + force_fb)
    frames = []
    for trial in range(24):
        cells = [(40 + 2 * (trial % 3), 30), (41 + 2 * (trial % 3), 30), (60, 20 + trial % 5), (60, 21 + trial % 5)]
        parts = []
        for ci, (cx, cy) in enumerate(cells):
            n = [vec - 1, vec, vec + 1, 4 * vec + 300][rng.randint(0, 4)]
            a = np.zeros(max(n, 0), dtype=m.MV_DTYPE)
            a["dst_x"], a["dst_y"] = 16 * cx + rng.randint(0, 16, size=len(a)), 16 * cy + rng.randint(0, 16, size=len(a))
            a["src_x"], a["src_y"] = a["dst_x"] - 7, a["dst_y"] + 3
            parts.append(a)
        fr = np.concatenate(parts)
        frames.append(fr[rng.permutation(len(fr))])
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert 0 < want.sum() < len(want)
    # ragged random frames with hot spots as well
    mv, off, sd = synth.random_frames(rng, 32, 4000, 1920, 1080, hot=0.7)
    assert_scan_parity(s, p, mv, off, sd)


# ------------------------------------------------------------------ scan: edge cases

EDGE_CFGS = [
    # (width, height, kwargs)
    (1920, 1080, dict()),
    (1920, 1080, dict(vertical_mask=0.0)),                              # margin 0: grid edges are centres
    (1920, 1080, dict(vectors_needed=1, clusters_needed=1)),
    (1920, 1080, dict(vectors_needed=0)),                               # every cell active
    (1920, 1080, dict(vectors_needed=255)),
    (1920, 1080, dict(vectors_needed=256 + 3)),                         # uint8 wrap -> 3
    (1920, 1080, dict(clusters_needed=0)), (1920, 1080, dict(clusters_needed=-5)),
    (1920, 1080, dict(clusters_needed=100000)),
    (1920, 1080, dict(mv_threshold_sq=0.0)), (1920, 1080, dict(mv_threshold_sq=-1.0)),
    (1920, 1080, dict(mv_threshold_sq=float("nan"))), (1920, 1080, dict(mv_threshold_sq=float("inf"))),
    (1920, 1080, dict(mv_threshold_sq=24.5)), (1920, 1080, dict(mv_threshold_sq=25.0)),
    (1920, 1080, dict(mv_threshold_sq=2.0e9)),
    (1920, 1080, dict(vertical_mask=0.5)), (1920, 1080, dict(vertical_mask=0.6)),   # empty analysed range
    (16, 16, dict(vertical_mask=0.0)), (32, 48, dict(vertical_mask=0.0)), (48, 48, dict(vertical_mask=0.0)),
    (1008, 64, dict(vertical_mask=0.0)), (1024, 64, dict(vertical_mask=0.0)),       # gw 63, 64
    (1040, 64, dict(vertical_mask=0.0)), (2064, 96, dict()),                         # gw 65, 129
    (1920, 1080, dict(block_size=8, block_shift=3)),                                 # 240x135 on 1080p
    (1920, 1080, dict(block_size=16, block_shift=5)),                                # size/shift mismatch
    (640, 480, dict(block_size=1, block_shift=0, vectors_needed=1)),                 # 640x480 cells
    (32767, 3, dict(block_size=1, block_shift=0, vectors_needed=1, vertical_mask=0.0)),   # widest legal grid
    (3, 32767, dict(block_size=1, block_shift=0, vectors_needed=1, vertical_mask=0.0)),   # tallest: chunked masks
    (32767, 40, dict(block_size=1, block_shift=0, vectors_needed=2, clusters_needed=1)),  # wide + row bands
]


@pytest.mark.parametrize("force_fb", [None, 8, 108])
@pytest.mark.parametrize("width,height,kw", EDGE_CFGS)
def test_scan_edge_configs(gpu_scanner_factory, width, height, kw, force_fb):
    import zlib
    rng = np.random.RandomState(zlib.crc32(repr((width, height, sorted(kw.items()))).encode()) % (2 ** 31))
    p = ob.params_from_config(width, height, **kw)
    assert m.ScanParams.from_config(width, height, **kw) == p or np.isnan(p.mv_threshold_sq)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    mv, off, sd = synth.random_frames(rng, 48, 3000, width, height)
    junk_padding(mv, rng)
    want = assert_scan_parity(s, p, mv, off, sd)
    # and with the NULL has_sd convention
    got = s.check_frames(m.FrameBatch(mv, off))
    assert np.array_equal(got, ob.scan_frames(p, mv, off, None))
    del want


BIG_D = [((65535, 65535), (65519, 65535)),      # 8 589 672 450 and 8 587 575 586: the largest sums int16 fields allow
         ((65535, 0), (65519, 0)),              # 4 294 836 225 (just below 2^32) and 4 292 739 361
         ((65535, 362), (65519, 2000)),         # 4 294 967 269 < 2^32 <= 4 296 739 361
         ((65535, 363), (65519, 2000)),         # 4 294 967 994 > 2^32: a sum truncated to 32 bits would read 698
         ((46341, 46341), (46340, 46341))]      # 4 294 976 562 and 4 294 883 881 on either side of 2^32


@pytest.mark.parametrize("thr", [4294836225.0, 4294836225.5, 4294967296.0, 4294967994.0, 4294967994.5, 8587575586.0,
                                 8587575586.5, 8589672450.0, 8589672450.5, 16.0])
def test_scan_magnitude_beyond_32_bits(gpu_scanner_factory, thr):
    """|d|^2 needs 34 bits when both differences span the whole int16 range.  Outside the reference's defined
    domain (signed overflow there); oracle and HIP path define it as the exact 64-bit magnitude.
    Two adjacent cells (the last two columns of a 2048 x 2048-cell grid, so dst = 32767 is inside it) each get one
    record; the frame fires iff BOTH magnitudes are >= the threshold (strict `<` of motion_scanner.cpp:251) —
    expected values from exact Python integers, thresholds on both sides of 2^32 and of the largest sums; through
    40-byte records and through the compact-record pipe (banded plan: 4 M cells)."""
    import math
    p = ob.params_from_config(32768, 32768, mv_threshold_sq=thr, vectors_needed=1, clusters_needed=1, vertical_mask=0.0)
    assert (p.grid_w, p.grid_h, p.vertical_margin) == (2048, 2048, 0)
    s = gpu_scanner_factory(p)
    frames, expect = [], []
    for (da, db) in BIG_D:
        mv = np.zeros(2, dtype=m.MV_DTYPE)
        mv["dst_x"], mv["dst_y"] = [32767, 32751], 32767                 # cells (2047, 2047) and (2046, 2047)
        mv["src_x"] = [32767 - da[0], 32751 - db[0]]
        mv["src_y"] = [32767 - da[1], 32767 - db[1]]
        frames.append(mv)
        expect.append(int(all(dx * dx + dy * dy >= math.ceil(thr) for dx, dy in (da, db))))
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert want.tolist() == expect
    pipe = m.ScanPipe(s, 16, 8, 2)
    for i, f in enumerate(frames):
        pipe.feed(f, float(i), tag=i)
    assert [fl for _, fl, _ in pipe.drain()] == expect
    pipe.close()


def test_scan_counter_saturation(gpu_scanner_factory):
    """> 255 votes in one cell (the reference's u8 saturates at 255, :265-266) and
    vectors_needed = 255: active iff >= 255 votes."""
    p = ob.params_from_config(1920, 1080, vectors_needed=255, clusters_needed=1)
    s = gpu_scanner_factory(p)
    frames = []
    for na, nb in [(254, 300), (255, 255), (1000, 70000), (255, 254)]:
        mv = np.zeros(na + nb, dtype=m.MV_DTYPE)
        mv["dst_x"][:na], mv["dst_x"][na:] = 800, 816
        mv["dst_y"] = 480
        mv["src_x"] = mv["dst_x"] + 10
        mv["src_y"] = mv["dst_y"]
        frames.append(mv)
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert list(want) == [0, 1, 1, 0]


def test_scan_empty_and_sideless_frames(gpu_scanner_factory):
    p0 = ob.params_from_config(1920, 1080, vectors_needed=0, clusters_needed=1)
    s0 = gpu_scanner_factory(p0)
    one = np.zeros(1, dtype=m.MV_DTYPE)
    b = m.FrameBatch.from_frames([None, np.zeros(0, dtype=m.MV_DTYPE), one, None])
    # vectors_needed == 0: "side data with zero records" -> phase 2 on an all-active grid -> true,
    # "no side data" -> false (reference :219-221 vs :282)
    want = assert_scan_parity(s0, p0, b.mv, b.frame_off, b.has_sd)
    assert list(want) == [0, 1, 1, 0]
    assert s0.check_frames(m.FrameBatch(b.mv, np.zeros(1, dtype=np.uint64))).size == 0


def test_scan_device_resident_matches_host_path(gpu_scanner_factory):
    import torch
    spec = synth.spec_1080p(seed=3, sub=1)
    spec.events = synth.scripted_events(spec, 64)
    mv, off, pts, sd = synth.gen_stream(spec, 64)
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    want = ob.scan_frames(p, mv, off, sd)
    assert want.sum() > 0
    d_mv = torch.from_numpy(mv.view(np.uint8).copy()).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    d_sd = torch.from_numpy(sd).cuda()
    got = s.check_frames_device(d_mv, d_off, d_sd)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        got2 = s.check_frames_device(d_mv, d_off, None)
    side.synchronize()
    assert np.array_equal(got2.cpu().numpy(), want)


@pytest.mark.parametrize("grid,cfg", [("1080p", "code_defaults"), ("4k", "code_defaults"), ("4k", "shipped_env")])
def test_scan_batch_properties_full_size(gpu_scanner_factory, grid, cfg):
    """Size-independent properties at bench scale (the oracle is too slow to recheck all of
    it): the flags of a tiled batch are the tile's flags repeated, and permuting frames
    permutes flags.  1080p and 4K (config 3, both parameter sets)."""
    import torch
    kw = dict(m.config.CODE_DEFAULTS if cfg == "code_defaults" else m.config.SHIPPED_ENV)
    if grid == "1080p":
        spec, (W, H), reps = synth.spec_1080p(seed=21), (1920, 1080), 128      # 4096 frames, 5.3 GB of records:
    else:                                                                        # offsets cross 2^32 bytes
        spec, (W, H), reps = synth.spec_4k(seed=22), (3840, 2160), 32          # 1024 4K frames, 5.3 GB
    spec.events = synth.scripted_events(spec, 32)
    mv, off, pts, sd = synth.gen_stream(spec, 32)
    p = ob.params_from_config(W, H, **kw)
    s = gpu_scanner_factory(p)
    want = ob.scan_frames(p, mv, off, None)
    assert 0 < want.sum() < 32
    d_tile = torch.from_numpy(mv.view(np.uint8).copy()).cuda()
    d_mv = d_tile.repeat(reps)
    counts = np.diff(off.astype(np.int64))
    off_big = np.concatenate([[0], np.cumsum(np.tile(counts, reps))]).astype(np.int64)
    got = s.check_frames_device(d_mv, torch.from_numpy(off_big).cuda()).cpu().numpy()
    assert np.array_equal(got, np.tile(want, reps))
    # reversed frame order: a reversed batch needs its own packed array; built on the host for the tile only
    order = np.arange(32)[::-1]
    frames = [mv[int(off[i]):int(off[i + 1])] for i in order]
    b = m.FrameBatch.from_frames(frames)
    assert np.array_equal(s.check_frames(m.FrameBatch(b.mv, b.frame_off)), want[order])


@pytest.mark.parametrize("group", ["1", ""])
@pytest.mark.parametrize("n_frames", [1, 63, 1024, 1025, 5000, 32768, 32769, 70001, 1100003])
def test_work_list_every_plan_size(gpu_scanner_factory, monkeypatch, n_frames, group):
    """The planning kernels at every size class: one block (<= 1024 frames), the fused form (<= 32 blocks: every block
    counts the frames before it again), the two-kernel form beyond, and more than 1024 x 1024 frames (every planning
    block then walks several times 1024 frames) — with MTGPU_GROUP=1 (one list entry per workgroup) and with the
    automatic grouping (from 4096 frames on, these tiny frames go two to eight consecutive ENTRIES to a workgroup,
    which parks the entries of its later frames in LDS when it starts).  Tiny frames (0..3
    records, a third of them without side data, some with EMPTY side data), so that the oracle checks every flag.  VECTORS_NEEDED 0
    makes the three kinds of frames differ: no side data -> false, empty side data -> true (every cell is active),
    records -> true."""
    import torch
    rng = np.random.RandomState(n_frames)
    kind = rng.randint(0, 6, size=n_frames)                   # 0, 1: no side data; 2: empty side data; 3..5: 1..3 records
    counts = np.where(kind >= 3, kind - 2, 0).astype(np.int64)
    has_sd = (kind >= 2).astype(np.uint8)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    n = int(off[-1])
    mv = np.zeros(max(n, 1), dtype=m.MV_DTYPE)[:n]
    mv["dst_x"], mv["dst_y"] = rng.randint(0, 1920, size=n), rng.randint(0, 1080, size=n)
    mv["src_x"], mv["src_y"] = mv["dst_x"] - rng.choice([0, 3], size=n), mv["dst_y"]      # |d|^2 <= 9: below the threshold
    three = off[:-1][kind == 5].astype(np.int64)              # frames with three records: two of them moving, side by side
    for q, gx in ((0, 50), (1, 51)):
        mv["dst_x"][three + q], mv["dst_y"][three + q], mv["src_x"][three + q], mv["src_y"][three + q] = gx * 16 + 8, 488, gx * 16, 488
    for vn, cn in ((0, 1), (1, 1)):
        p = ob.params_from_config(1920, 1080, vectors_needed=vn, clusters_needed=cn)
        if group:
            monkeypatch.setenv("MTGPU_GROUP", group)
        s = gpu_scanner_factory(p)
        monkeypatch.delenv("MTGPU_GROUP", raising=False)
        want = ob.scan_frames(p, mv, off, has_sd, nthreads=8)
        got = s.check_frames(m.FrameBatch(mv, off, None, has_sd))
        bad = np.flatnonzero(want != got)
        assert bad.size == 0, (vn, bad[:8], want[bad[:8]], got[bad[:8]], kind[bad[:8]])
        if vn == 0:
            assert np.array_equal(want, has_sd)              # side data (even empty) -> every cell active -> true
        else:
            assert np.array_equal(want, (kind == 5).astype(np.uint8))
        # device entry point, has_sd == NULL (side data iff records), poisoned flags: every byte is written
        d_flags = torch.full((n_frames,), 9, dtype=torch.uint8, device="cuda")
        d_mv = torch.from_numpy(mv.view(np.uint8).copy()).cuda() if n else torch.zeros(40, dtype=torch.uint8, device="cuda")
        s.check_frames_device(d_mv[: n * 40], torch.from_numpy(off.astype(np.int64)).cuda(), None, d_flags)
        want2 = ob.scan_frames(p, mv, off, None, nthreads=8)
        assert np.array_equal(d_flags.cpu().numpy(), want2), vn


# ------------------------------------------------------------------ merge

def merge_case(scanner, ts, mp, job):
    want_seg, want_res = ob.pool_and_merge(ts, mp, job)
    got_seg, got_res = scanner.merge_segments(ts, mp, job)
    assert got_res["n_timestamps"] == want_res["n_timestamps"]
    assert got_res["n_segments"] == want_res["n_segments"]
    assert got_res["do_cut"] == want_res["do_cut"]
    for key in ("time_removed", "saved_pct"):        # bit-exact; a NaN (inf/inf with an infinite duration) only has to be a NaN:
        g, w = got_res[key], want_res[key]           # x86 SSE and the GPU produce default NaNs of opposite sign
        assert (g != g and w != w) or bits([g]).tolist() == bits([w]).tolist(), (key, g, w)
    assert np.array_equal(bits(got_seg["start"]), bits(want_seg["start"]))
    assert np.array_equal(bits(got_seg["end"]), bits(want_seg["end"]))
    return want_seg, want_res


@pytest.mark.parametrize("job", [False, True])
def test_merge_random(gpu_scanner_factory, job):
    s = gpu_scanner_factory(ob.params_from_config(1920, 1080))
    rng = np.random.RandomState(5)
    for n in [0, 1, 2, 3, 17, 64, 65, 1023, 1024, 1025, 5000, 40000]:
        dur = 600.0
        base = np.sort(rng.rand(n) * dur)
        # cluster the timestamps so gaps fall on both sides of MAX_GAP_SEC
        ts = np.round(base / 7.0) * 7.0 + rng.rand(n) * 3.0 if n else base
        for variant in ("sorted", "shuffled", "dups"):
            v = np.sort(ts)
            if variant == "shuffled":
                v = rng.permutation(v)
            if variant == "dups" and n:
                v = rng.permutation(np.concatenate([v, v[: n // 2 + 1], v[:1]]))
            for mp in (m.MergeParams(duration=dur, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0),
                       m.MergeParams(duration=dur * 0.5, max_gap_sec=0.25, padding_sec=2.0, min_savings_pct=5.0),
                       m.MergeParams(duration=0.0, max_gap_sec=1.0, padding_sec=0.0, min_savings_pct=5.0)):
                merge_case(s, v, mp, job)


def test_merge_frame_timestamps_and_capacity(gpu_scanner_factory):
    s = gpu_scanner_factory(ob.params_from_config(1920, 1080))
    spec = synth.spec_1080p()
    idx = np.r_[301:720, 1801:1830, 4000:4002]
    ts = np.array([spec.pts_seconds(i) for i in idx])
    mp = m.MergeParams(duration=150.0, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0)
    seg, res = merge_case(s, ts, mp, True)
    assert res["n_segments"] == 3 and res["do_cut"] == 1
    # capacity: fewer slots than segments -> MT_ERR_CAPACITY, n_segments still reported
    with pytest.raises(m.MtgpuError) as ei:
        s.merge_segments(ts, mp, False, cap=2)
    assert ei.value.code == 2
    with pytest.raises(m.MtgpuError) as ei:
        s.merge_segments(np.array([1.0, float("nan")]), mp)
    assert ei.value.code == 1


def test_merge_streams_on_device(gpu_scanner_factory):
    import torch
    p = ob.params_from_config(1920, 1080)
    s = gpu_scanner_factory(p)
    rng = np.random.RandomState(8)
    n_streams = 9
    lens = [0, 1, 700, 2048, 33, 5000, 1024, 1, 300]
    flags, pts, mps, off = [], [], [], [0]
    for i, n in enumerate(lens):
        f = (rng.rand(n) < 0.3).astype(np.uint8)
        # bursts
        for _ in range(3):
            if n > 10:
                a = rng.randint(0, n - 5)
                f[a:a + rng.randint(1, 60)] = 1
        if i == 7:
            f[:] = 0
        t = np.arange(n) * (1.0 / 30.0) + i
        if i == 4:
            t = rng.permutation(t)                    # out-of-order stream (pooled chunks)
        if i == 8:
            t = np.repeat(t[::3], 3)[:n]              # duplicates
        flags.append(f)
        pts.append(t)
        off.append(off[-1] + n)
        mps.append(m.MergeParams(duration=float(n) / 30.0 + i, max_gap_sec=1.0 + 0.5 * i,
                                 padding_sec=0.25 * i, min_savings_pct=5.0))
    flags = np.concatenate(flags)
    pts = np.concatenate(pts)
    mp_rec = np.concatenate([x.to_record() for x in mps])
    for job in (False, True):
        seg, res = s.merge_streams_device(
            torch.from_numpy(flags).cuda(), torch.from_numpy(pts).cuda(),
            torch.from_numpy(np.array(off, dtype=np.int64)).cuda(),
            torch.from_numpy(mp_rec.view(np.uint8).copy()).cuda(), job_semantics=job, seg_cap=4096)
        torch.cuda.synchronize()
        seg = seg.cpu().numpy()
        res = m.results_from_bytes(res.cpu().numpy())
        for i in range(n_streams):
            a, b = off[i], off[i + 1]
            want_seg, want_res = ob.pool_and_merge(pts[a:b][flags[a:b] != 0], mps[i], job)
            assert res[i]["status"] == 0
            assert res[i]["n_timestamps"] == want_res["n_timestamps"], i
            assert res[i]["n_segments"] == want_res["n_segments"], i
            assert res[i]["do_cut"] == want_res["do_cut"], i
            k = int(res[i]["n_segments"])
            assert np.array_equal(bits(seg[i, :k, 0]), bits(want_seg["start"])), i
            assert np.array_equal(bits(seg[i, :k, 1]), bits(want_seg["end"])), i
            assert bits([res[i]["time_removed"], res[i]["saved_pct"]]).tolist() == \
                bits([want_res["time_removed"], want_res["saved_pct"]]).tolist()


def test_end_to_end_stream_segments(gpu_scanner_factory):
    """Synthetic 1080p stream -> flags -> timestamps -> segments, GPU vs oracle, through
    the reference-shaped host interface (scan_range per chunk, pooled, merged)."""
    spec = synth.spec_1080p(seed=33, sub=1)
    n = 600                                           # 20 s at 30 fps
    spec.events = synth.scripted_events(spec, n)
    frames = [synth.gen_frame(spec, i) for i in range(n)]
    ticks = [spec.pts_ticks(i) for i in range(n)]
    tb = 1.0 / spec.tb_den
    duration = n / spec.fps
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    pooled_gpu, pooled_cpu = [], []
    for (c0, c1, _) in m.make_chunks(duration, 6.0):
        # a decoder seeks back to the previous keyframe: hand over frames from one GOP earlier
        first = max(0, int(c0 * spec.fps) - spec.gop)
        sub_ticks, sub_frames = ticks[first:], frames[first:]
        pooled_gpu.append(s.scan_range(sub_ticks, sub_frames, tb, c0, c1, spec.fps, target_fps=10.0))
        idx, pts = ob.filter_frames(sub_ticks, tb, c0, c1, m.frame_skip(spec.fps, 10.0))
        b = m.FrameBatch.from_frames([sub_frames[i] for i in idx])
        fl = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
        pooled_cpu.append([t for t, f in zip(pts, fl) if f])
    assert pooled_gpu == pooled_cpu
    ts = [t for c in reversed(pooled_gpu) for t in c]  # workers finish in any order
    assert len(ts) > 10
    mp = m.MergeParams(duration=duration)
    merge_case(s, np.array(ts), mp, True)


# ------------------------------------------------------------------ host dispatcher (pipe) + C++ host layer

def test_scan_pipe_matches_oracle(gpu_scanner_factory):
    """Pinned multi-buffered pipe: frames fed one by one (as a decoder thread would), small
    batches so that capacity splits, back-pressure and partial batches all occur."""
    spec = synth.spec_1080p(seed=17, sub=1)
    spec.events = synth.scripted_events(spec, 150)
    frames = [synth.gen_frame(spec, i) for i in range(150)]
    frames[7] = np.zeros(0, dtype=m.MV_DTYPE)                    # side data with zero records
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    b = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    for (max_rec, max_fr, nbuf) in [(8160 * 5, 7, 2), (8160 * 64, 64, 3), (8160, 1, 1), (8160 * 3 + 17, 1000, 4)]:
        pipe = m.ScanPipe(s, max_rec, max_fr, nbuf)
        for i, f in enumerate(frames):
            pipe.feed(f, spec.pts_seconds(i), tag=i)
        out = pipe.drain()
        assert [t for _, _, t in out] == list(range(150))        # submission order
        assert [fl for _, fl, _ in out] == want.tolist()
        assert [pt for pt, _, _ in out] == [spec.pts_seconds(i) for i in range(150)]
        assert pipe.drain() == []
        pipe.close()
    pipe = m.ScanPipe(s, 100, 4, 2)                               # a frame larger than a whole batch:
    pipe.feed(frames[1], 0.0)                                     # the empty batch grows (check_frame takes any count)
    assert [fl for _, fl, _ in pipe.drain()] == [int(want[1])]
    pipe.close()


def _reference_worker_loop(spec, frames, ticks, duration, p, chunk_sec, target_fps, mp):
    """Python transcription of the reference's worker loop + merge, driven by the oracle:
    chunks -> backward seek to a keyframe -> frame filter -> check_frame -> pool -> merge."""
    tb = 1.0 / spec.tb_den
    skip = m.frame_skip(spec.fps, target_fps)
    pooled = []
    keys = [i for i, f in enumerate(frames) if f is None]
    for (c0, c1, _) in m.make_chunks(duration, chunk_sec):
        target = int(c0 / tb)                                     # motion_scanner.cpp:322
        first = max([k for k in keys if ticks[k] <= target] or [0])
        idx, pts = ob.filter_frames(ticks[first:], tb, c0, c1, skip)
        bt = m.FrameBatch.from_frames([frames[first + i] for i in idx])
        fl = ob.scan_frames(p, bt.mv, bt.frame_off, bt.has_sd)
        pooled += [t for t, f in zip(pts, fl) if f]
    return pooled, ob.pool_and_merge(pooled, mp, True)


def _check_job(r, pooled, want_seg, want_res):
    assert r["motion_frames"] == len(pooled)
    assert r["n_timestamps"] == want_res["n_timestamps"] and r["do_cut"] == want_res["do_cut"]
    assert [[float(a).hex(), float(b).hex()] for a, b in r["segments"]] == \
        [[float(a).hex(), float(b).hex()] for a, b in want_seg.tolist()]
    assert float(r["time_removed"]).hex() == float(want_res["time_removed"]).hex()
    assert float(r["saved_pct"]).hex() == float(want_res["saved_pct"]).hex()


def _make_stream_file(tmp_path, name, seed, n, still=False):
    spec = synth.StreamSpec(width=640, height=480, block=16, sub=1, fps=30.0, gop=15, seed=seed,
                            salt_p=0.0 if still else 1e-3)
    spec.events = [] if still else synth.scripted_events(spec, n)
    frames = [synth.gen_frame(spec, i) for i in range(n)]
    ticks = [spec.pts_ticks(i) for i in range(n)]
    path = str(tmp_path / name)
    m.mvfile.write_mtmv(path, 640, 480, 1, spec.tb_den, spec.fps, n / spec.fps, ticks, frames)
    return spec, frames, ticks, path


def test_cpp_host_pipeline_scan_file(tmp_path):
    """The C++ host layer (csrc/host/mtgpu_host.hpp: TaskQueue, ResultCollector,
    GpuMotionScanner::scan_range, run_scan_pipeline, process_batch) through its front end
    mtgpu_scan_file on .mtmv streams, against a Python transcription of the reference's worker
    loop driven by the oracle."""
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(m.LIB_PATH), "mtgpu_scan_file")
    assert os.path.exists(exe), "build it with make -C motion-estimated-video-trimmer_amd/csrc"
    env_cfg = dict(VECTORS_NEEDED="1", CHUNK_DURATION_SEC="4", TARGET_FPS="10", MAX_GAP_SEC="2.0",
                   PADDING_SEC="0.5", MIN_SAVINGS_PCT="5")
    env = dict(os.environ, **env_cfg)
    p = ob.params_from_config(640, 480, vectors_needed=1)

    # ---- one video: ProcessingPipeline-shaped run, any worker count gives the same job
    n = 900                                                       # 30 s
    spec, frames, ticks, path = _make_stream_file(tmp_path, "s.mtmv", 5, n)
    mp = m.MergeParams(duration=n / spec.fps, max_gap_sec=2.0, padding_sec=0.5, min_savings_pct=5.0)
    pooled, (want_seg, want_res) = _reference_worker_loop(spec, frames, ticks, n / spec.fps, p, 4.0, 10.0, mp)
    assert len(want_seg) >= 2
    for threads in (1, 3, 8):
        out = subprocess.run([exe, path, "--threads", str(threads)], check=True, capture_output=True,
                             text=True, env=env).stdout
        r = json.loads(out)
        assert r["chunks"] == 8
        _check_job(r, pooled, want_seg, want_res)

    # ---- a batch of videos: BatchProcessor-shaped run (S streams x T workers, one job consumer)
    cases = {path: (pooled, want_seg, want_res)}
    paths = [path]
    for k, (seed, nn, still) in enumerate([(21, 450, False), (22, 600, False), (23, 300, True), (24, 750, False)]):
        sp, fr, tk, pth = _make_stream_file(tmp_path, f"b{k}.mtmv", seed, nn, still)
        mpk = m.MergeParams(duration=nn / sp.fps, max_gap_sec=2.0, padding_sec=0.5, min_savings_pct=5.0)
        pl, (ws, wr) = _reference_worker_loop(sp, fr, tk, nn / sp.fps, p, 4.0, 10.0, mpk)
        cases[pth] = (pl, ws, wr)
        paths.append(pth)
    assert cases[paths[3]][2]["do_cut"] == -1                     # the still video: no motion, no job
    for streams, threads in ((1, 2), (3, 2), (5, 1)):
        out = subprocess.run([exe] + paths + ["--streams", str(streams), "--threads", str(threads),
                                              "--outdir", str(tmp_path)], check=True, capture_output=True,
                             text=True, env=env).stdout
        jobs = [json.loads(ln) for ln in out.strip().splitlines()]
        assert sorted(j["input"] for j in jobs) == sorted(pth for pth in paths if cases[pth][2]["do_cut"] >= 0)
        for j in jobs:
            _check_job(j, *cases[j["input"]])


def test_scan_device_unaligned_base_and_offset_clamp(gpu_scanner_factory):
    """The record array may start at any 4-byte boundary (sub-views of a larger buffer), and
    frame offsets beyond n_records are clamped instead of read (a bad offset must not fault)."""
    import torch
    rng = np.random.RandomState(4)
    p = ob.params_from_config(1920, 1080, vectors_needed=1, clusters_needed=1)
    s = gpu_scanner_factory(p)
    mv, off, sd = synth.random_frames(rng, 40, 2500, 1920, 1080)
    want = ob.scan_frames(p, mv, off, None)
    raw = torch.from_numpy(mv.view(np.uint8).copy())
    for shift in (4, 8, 12, 20):
        buf = torch.zeros(raw.numel() + 64, dtype=torch.uint8, device="cuda")
        view = buf[shift:shift + raw.numel()]
        view.copy_(raw)
        assert view.data_ptr() % 16 == shift % 16
        got = s.check_frames_device(view, torch.from_numpy(off.astype(np.int64)).cuda())
        assert np.array_equal(got.cpu().numpy(), want), shift
    # clamp: pretend the array holds only the records of the first 25 frames
    n_keep = int(off[25])
    d_mv = raw[: n_keep * 40].cuda()
    got = s.check_frames_device(d_mv, torch.from_numpy(off.astype(np.int64)).cuda()).cpu().numpy()
    off_clamped = np.minimum(off, n_keep)
    assert np.array_equal(got, ob.scan_frames(p, mv[:n_keep], off_clamped, None))
    # an unaligned (odd) base is rejected, not dereferenced
    bad = torch.zeros(raw.numel() + 8, dtype=torch.uint8, device="cuda")[1:1 + raw.numel()]
    with pytest.raises(m.MtgpuError) as ei:
        s.check_frames_device(bad, torch.from_numpy(off.astype(np.int64)).cuda())
    assert ei.value.code == 1


def test_c_abi_gather_single_rank(gpu_scanner_factory):
    """mtgpu_comm_* / mtgpu_gather_segments (RCCL wrappers of the C ABI) with a 1-rank
    communicator: the packed segment block comes back unchanged.  (N > 1 needs N GPUs; the
    N-rank packing/assembly logic is covered by tests/test_dist_gloo.py.)"""
    import ctypes as C
    import torch
    from mvtrim_amd import dist as mdist
    lib = m.load_library()
    uid = (C.c_char * 128)()
    m._abi.check(lib.mtgpu_comm_unique_id(uid))
    comm = C.c_void_p()
    m._abi.check(lib.mtgpu_comm_create(0, 1, uid, 0, C.byref(comm)))
    try:
        s = gpu_scanner_factory(ob.params_from_config(1920, 1080))
        flags = torch.from_numpy((np.arange(600) % 97 < 20).astype(np.uint8)).cuda()
        pts = torch.from_numpy(np.arange(600) / 30.0).cuda()
        soff = torch.tensor([0, 200, 600], dtype=torch.int64, device="cuda")
        mp = np.concatenate([m.MergeParams(duration=200 / 30.0, max_gap_sec=1.0).to_record(),
                             m.MergeParams(duration=400 / 30.0, max_gap_sec=1.0).to_record()])
        seg, res = s.merge_streams_device(flags, pts, soff, torch.from_numpy(mp.view(np.uint8).copy()).cuda(),
                                          True, 16)
        packed = mdist.pack_segment_lists(seg, res)
        recv = torch.zeros((1,) + tuple(packed.shape), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        m._abi.check(lib.mtgpu_gather_segments(comm, packed.data_ptr(), packed.numel(), recv.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(recv[0], packed)
        lists = mdist.assemble_stream_lists(recv, 16, [2])
        assert len(lists) == 2 and int(lists[0]["result"]["n_segments"]) >= 1
    finally:
        lib.mtgpu_comm_destroy(comm)


def test_scan_host_window_of_larger_batch(gpu_scanner_factory):
    """frame_off need not start at 0: scanning frames [a, b) of a larger packed array copies
    only that window to the device (mtgpu_scan_frames rebases the record base)."""
    rng = np.random.RandomState(12)
    p = ob.params_from_config(1920, 1080, vectors_needed=1, clusters_needed=1)
    s = gpu_scanner_factory(p)
    mv, off, sd = synth.random_frames(rng, 60, 2000, 1920, 1080)
    want = ob.scan_frames(p, mv, off, sd)
    for a, b in [(0, 60), (10, 35), (59, 60), (17, 17)]:
        got = s.check_frames(m.FrameBatch(mv, off[a:b + 1], None, sd[a:b]))
        assert np.array_equal(got, want[a:b]), (a, b)


def test_scan_fuzz_random_configs(gpu_scanner_factory):
    """Randomised sweep: grid sizes, block settings, thresholds, vote / cluster needs, masks and
    every counter form, each on ragged adversarial frames.  Bit-exact flags vs the oracle."""
    rng = np.random.RandomState(20260104)
    forms = [None, 1, 2, 4, 8, 108, 32]
    n_cfg = 120
    for it in range(n_cfg):
        sh = int(rng.randint(0, 7))
        bs = int(rng.choice([1 << sh, 16, 8]))
        w, h = int(rng.randint(8, 4300)), int(rng.randint(8, 2400))
        kw = dict(mv_threshold_sq=float(rng.choice([16.0, 4.0, 0.0, 1.0, 30.25, 1e4, float("nan")])),
                  block_size=bs, block_shift=sh,
                  vectors_needed=int(rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 17, 255, 256 + 2])),
                  clusters_needed=int(rng.choice([-1, 0, 1, 2, 3, 7, 50])),
                  vertical_mask=float(rng.choice([0.0, 0.05, 0.1, 0.3, 0.49, 0.5])))
        try:
            p = ob.params_from_config(w, h, **kw)
        except ValueError:
            continue
        if p.grid_w * p.grid_h > 1_500_000:         # keep the oracle fast
            continue
        try:
            s = gpu_scanner_factory(p, force_fb=forms[it % len(forms)])
        except m.MtgpuError as e:
            assert e.code == 2, e                    # only MT_ERR_CAPACITY (grid row wider than LDS) is acceptable
            continue
        mv, off, sd = synth.random_frames(rng, 10, 2500, w, h, hot=float(rng.choice([0.1, 0.5, 0.9])))
        junk_padding(mv, rng)
        want = ob.scan_frames(p, mv, off, sd)
        got = s.check_frames(m.FrameBatch(mv, off, None, sd))
        assert np.array_equal(got, want), (it, w, h, kw, s.plan)
        s.close()


@pytest.mark.parametrize("slices", [2, 4, 8])
@pytest.mark.parametrize("force_fb", [None, 1, 2, 4, 8, 108])
def test_scan_frame_slices(gpu_scanner_factory, slices, force_fb):
    """Several workgroups per frame (record-array slices, last arriver sums the partial LDS tiles):
    results must not depend on the slice count, for every counter form."""
    rng = np.random.RandomState(slices * 100 + (force_fb or 0))
    for vec, clus in ((1, 1), (2, 2), (4, 3), (8, 2), (20, 1)):
        if force_fb not in (None, 108) and vec > force_fb:
            continue
        p = ob.params_from_config(1920, 1080, vectors_needed=vec, clusters_needed=clus)
        s = gpu_scanner_factory(p, force_fb=force_fb)
        s.set_slices(slices)
        # cells whose votes are split across slices: sitting at vec-1 / vec / above only in the SUM
        frames = []
        for trial in range(12):
            parts = []
            for (cx, cy) in [(30, 20), (31, 20), (70, 40), (70, 41)]:
                n = [vec - 1, vec, vec + 1, 3 * vec + 40][rng.randint(0, 4)]
                a = np.zeros(max(n, 0), dtype=m.MV_DTYPE)
                a["dst_x"], a["dst_y"] = 16 * cx + 3, 16 * cy + 5
                a["src_x"], a["src_y"] = a["dst_x"] - 9, a["dst_y"]
                parts.append(a)
            filler = np.zeros(rng.randint(0, 400), dtype=m.MV_DTYPE)      # below threshold
            filler["dst_x"], filler["dst_y"] = rng.randint(0, 1920, size=len(filler)), rng.randint(0, 1080, size=len(filler))
            filler["src_x"], filler["src_y"] = filler["dst_x"] + 1, filler["dst_y"]
            fr = np.concatenate(parts + [filler])
            frames.append(fr[rng.permutation(len(fr))])               # votes of one cell land in different slices
        frames += [None, np.zeros(0, dtype=m.MV_DTYPE), frames[0][:3].copy()]
        b = m.FrameBatch.from_frames(frames)
        want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
        assert 0 < want.sum() < len(want)
        mv, off, sd = synth.random_frames(rng, 24, 6000, 1920, 1080, hot=0.6)
        assert_scan_parity(s, p, mv, off, sd)
        s.close()


def test_scan_frame_slices_big_grids_and_auto(gpu_scanner_factory):
    import torch
    # 4K u32 tile, sliced
    spec = synth.spec_4k(seed=5)
    spec.events = [synth.Event(1, 6, 100, 60, 4, 3, 9, -5)]
    mv, off, pts, sd = synth.gen_stream(spec, 8)
    p = ob.params_from_config(3840, 2160)
    want = ob.scan_frames(p, mv, off, sd)
    for S in (0, 1, 4, 8):                                          # 0 = auto (8 frames << 512 slots: slices)
        s = gpu_scanner_factory(p)
        s.set_slices(S)
        assert np.array_equal(s.check_frames(m.FrameBatch(mv, off, None, sd)), want), S
    # 960x540 grid, 1-bit and 2-bit thermometer tiles, sliced; device-resident entry point
    specf = synth.spec_4k_fine(seed=9)
    specf.events = [synth.Event(1, 3, 300, 200, 6, 6, 9, 3), synth.Event(1, 4, 100, 250, 3, 120, 9, 0)]
    mvf, offf, ptsf, sdf = synth.gen_stream(specf, 4)
    for vec, fb in ((1, None), (1, 2), (2, None)):
        pf = ob.params_from_config(3840, 2160, block_size=4, block_shift=2, vectors_needed=vec)
        wantf = ob.scan_frames(pf, mvf, offf, sdf)
        for S in (0, 2, 8):
            s = gpu_scanner_factory(pf, force_fb=fb)
            s.set_slices(S)
            got = s.check_frames_device(torch.from_numpy(mvf.view(np.uint8).copy()).cuda(),
                                        torch.from_numpy(offf.astype(np.int64)).cuda(),
                                        torch.from_numpy(sdf).cuda())
            assert np.array_equal(got.cpu().numpy(), wantf), (vec, fb, S)
    with pytest.raises(m.MtgpuError):
        s.set_slices(3)


def test_timerange_sharded_stream_single_rank_group(gpu_scanner_factory):
    """dist.scan_and_merge_timerange with a 1-rank NCCL(RCCL) group: scan -> compaction ->
    all_gather of timestamps -> device merge, against the oracle.  (The 2-rank exchange itself
    is covered on CPU by tests/test_dist_gloo.py.)"""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from mvtrim_amd import dist as mdist
    spec = synth.spec_1080p(seed=41, sub=1)
    n = 300
    spec.events = [synth.Event(20, 70, 30, 20, 4, 3, 9, 1), synth.Event(150, 230, 60, 40, 3, 3, -8, 2)]
    mv, off, pts, sd = synth.gen_stream(spec, n)
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    mp = m.MergeParams(duration=n / 30.0, max_gap_sec=1.0, padding_sec=0.25, min_savings_pct=5.0)
    want_flags = ob.scan_frames(p, mv, off, sd)
    want_seg, want_res = ob.pool_and_merge(pts[want_flags != 0], mp, True)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        seg, rec = mdist.scan_and_merge_timerange(
            s, torch.from_numpy(mv.view(np.uint8).copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(),
            torch.from_numpy(pts).cuda(), mp, has_sd=torch.from_numpy(sd).cuda())
    finally:
        dist.destroy_process_group()
    assert int(rec["do_cut"]) == want_res["do_cut"] and int(rec["n_timestamps"]) == want_res["n_timestamps"]
    assert seg.tobytes() == np.stack([want_seg["start"], want_seg["end"]], 1).tobytes()
    assert float(rec["time_removed"]) == want_res["time_removed"]


def test_scan_frame_slices_under_load(gpu_scanner_factory):
    """Slice hand-off under full load with a warm L1 and a reused workspace: thousands of frames
    x 8 slices keep every CU busy while tiles are published and summed; repeated launches reuse
    the same stream-ordered workspace addresses, so a missing acquire would read stale tiles.
    Flags must equal the unsliced kernel's (which equal the oracle's on the tile)."""
    import torch
    rng = np.random.RandomState(3)
    p = ob.params_from_config(1920, 1080, vectors_needed=2, clusters_needed=2)
    tiles = []
    for t in range(3):                                        # three different batches, same shapes
        spec = synth.spec_1080p(seed=50 + t, sub=2)
        spec.events = synth.scripted_events(spec, 24, seed=t)
        mv, off, pts, sd = synth.gen_stream(spec, 24)
        tiles.append((mv, off, ob.scan_frames(p, mv, off, None)))
    s1 = gpu_scanner_factory(p)
    s1.set_slices(1)
    s8 = gpu_scanner_factory(p)
    s8.set_slices(8)
    s2 = gpu_scanner_factory(p, force_fb=2)
    s2.set_slices(4)
    reps = 80                                                 # 1920 frames, 15 360 workgroups with 8 slices
    for rnd in range(3):
        for (mv, off, want) in tiles:
            d_mv = torch.from_numpy(mv.view(np.uint8).copy()).cuda().repeat(reps)
            counts = np.tile(np.diff(off.astype(np.int64)), reps)
            d_off = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).cuda()
            big = np.tile(want, reps)
            for sc in (s8, s1, s2, s8):
                got = sc.check_frames_device(d_mv, d_off).cpu().numpy()
                assert np.array_equal(got, big), (rnd, sc.plan)
    del rng


def test_scan_launch_chunking(gpu_scanner_factory):
    """Batches above 2^30 work items are scanned in several launches (grid.x < 2^31); the
    MTGPU_ITEM_CHUNK knob shrinks the launch size so that the item-offset path is exercised,
    also together with slices and row bands."""
    import os
    rng = np.random.RandomState(77)
    cases = [(ob.params_from_config(1920, 1080, vectors_needed=1, clusters_needed=1), None, 0),
             (ob.params_from_config(1920, 1080, vectors_needed=1, clusters_needed=1), None, 4),
             (ob.params_from_config(3840, 2160, block_size=4, block_shift=2, vectors_needed=1), 32, 0)]   # row bands
    for p, fb, slices in cases:
        w = 1920 if p.grid_w == 120 else 3840
        h = 1080 if p.grid_w == 120 else 2160
        mv, off, sd = synth.random_frames(rng, 45, 1500, w, h)
        want = ob.scan_frames(p, mv, off, sd)
        os.environ["MTGPU_ITEM_CHUNK"] = "7"
        try:
            s = gpu_scanner_factory(p, force_fb=fb)
        finally:
            del os.environ["MTGPU_ITEM_CHUNK"]
        if slices:
            s.set_slices(slices)
        got = s.check_frames(m.FrameBatch(mv, off, None, sd))
        assert np.array_equal(got, want), (fb, slices, s.plan)


def test_scan_frames_per_workgroup_grouping(gpu_scanner_factory, monkeypatch):
    """Small frames are scanned several per workgroup (MTGPU_GROUP forces a count): the flags must
    not depend on the grouping, including ragged tails, frames without side data inside a group,
    banded plans and the compact layout."""
    import torch
    rng = np.random.RandomState(31)
    cases = [(1920, 1080, dict(vectors_needed=1, clusters_needed=1), None),
             (3840, 2160, dict(block_size=4, block_shift=2, vectors_needed=4), None)]       # row bands (spill queue)
    for w, h, kw, fb in cases:
        p = ob.params_from_config(w, h, **kw)
        mv, off, sd = synth.random_frames(rng, 53, 1800, w, h, hot=0.4)
        want = ob.scan_frames(p, mv, off, sd)
        for g in (1, 2, 3, 7, 64):
            monkeypatch.setenv("MTGPU_GROUP", str(g))
            s = gpu_scanner_factory(m.ScanParams.from_config(w, h, **kw), force_fb=fb)
            monkeypatch.delenv("MTGPU_GROUP")
            assert np.array_equal(s.check_frames(m.FrameBatch(mv, off, None, sd)), want), (w, g)
            rec = m.pack_records(mv)
            got = s.check_frames_device_compact(torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda(),
                                                torch.from_numpy(off.astype(np.int64)).cuda(),
                                                torch.from_numpy(sd).cuda()).cpu().numpy()
            assert np.array_equal(got, want), (w, g, "compact")


def test_scan_line_aligned_streams(gpu_scanner_factory, monkeypatch):
    """The record streams start on a 128-byte line: up to 15 head records of every frame are scanned one by one
    (scan_kernels.hip, "Line alignment").  Frames whose first record sits at every possible offset inside a line
    (40-byte records: 16 residues; compact: 16), frames shorter than their head, with the alignment on and off,
    single-tile and banded plans, and voters placed exactly in the head and right behind it."""
    import torch
    rng = np.random.RandomState(5)
    p = ob.params_from_config(1920, 1080, vectors_needed=2, clusters_needed=1)
    frames, kinds = [], []
    for i, n in enumerate([3, 0, 5, 1, 7, 2600, 2, 9, 11, 4, 13, 6, 1, 15, 8, 2601, 10, 12, 14, 16, 3000, 1, 1, 1, 17, 2, 2, 5000]):
        mv = np.zeros(n, dtype=m.MV_DTYPE)
        mv["dst_x"] = rng.randint(0, 1920, size=n)
        mv["dst_y"] = rng.randint(0, 1080, size=n)
        mv["src_x"], mv["src_y"] = mv["dst_x"] - 1, mv["dst_y"]
        kind = "-"
        if n >= 4:
            kind = "AB"[i % 2]
            cx, cy = int(rng.randint(2, 110)), int(rng.randint(6, 60))
            pos = sorted({0, 1, min(n - 1, 15), n - 1} | ({16, 17} if n > 20 else set()))[:4]
            while len(pos) < 4:
                pos = sorted(set(pos) | {int(rng.randint(0, n))})
            cells = [(cx, cy), (cx + 1, cy), (cx, cy), (cx + 1, cy)]
            if kind == "B":
                cells[3] = (cx + 7, cy + 2)                        # the second cell keeps one vote: inactive
            for q, (gx, gy) in zip(pos, cells):
                mv["dst_x"][q], mv["dst_y"][q] = gx * 16 + 8, gy * 16 + 8
                mv["src_x"][q], mv["src_y"][q] = gx * 16 + 2, gy * 16 + 8
        frames.append(mv)
        kinds.append(kind)
    b = m.FrameBatch.from_frames(frames)
    assert len({(int(o) * 40) % 128 for o in b.frame_off[:-1]}) >= 12 and len({(int(o) * 8) % 128 for o in b.frame_off[:-1]}) >= 12
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    assert want.tolist() == [1 if k == "A" else 0 for k in kinds]
    rec = m.pack_records(b.mv)
    d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda()
    d_off = torch.from_numpy(b.frame_off.astype(np.int64)).cuda()
    for align in (("1", "0") if experiments_build() else ("1",)):     # MTGPU_ALIGN=0: experiments build only
        for fb in (None, 2, 8):                                     # 32-bit single tile, packed, banded (spill queue)
            monkeypatch.setenv("MTGPU_ALIGN", align)
            s = gpu_scanner_factory(m.ScanParams.from_config(1920, 1080, vectors_needed=2, clusters_needed=1), force_fb=fb)
            monkeypatch.delenv("MTGPU_ALIGN")
            assert np.array_equal(s.check_frames(b), want), (align, fb)
            assert np.array_equal(s.check_frames_device_compact(d_rec, d_off, None).cpu().numpy(), want), (align, fb, "compact")


@pytest.mark.parametrize("grid", ["1080p", "4k"])
def test_compact_next_frame_prefetch(gpu_scanner_factory, monkeypatch, grid):
    """Compact records, several frames per workgroup: the first streaming step of the next frame is issued before
    this frame's cluster test (scan_kernels.hip, NextStep).  Frame sizes on both sides of one full step
    (4 x BLOCK pairs), odd record counts (the next frame then starts 8 bytes off a 16-byte boundary), empty
    frames, frames without side data and I-frame gaps inside a group, a group's last frame, batches that end
    inside a group; with the prefetch on and off, the flags must be the oracle's."""
    import torch
    w, h = (1920, 1080) if grid == "1080p" else (3840, 2160)
    block = 512 if grid == "1080p" else 1024
    step = 2 * 4 * block                                 # records of one full streaming step (4 pairs per lane)
    rng = np.random.RandomState(77 + block)
    sizes = [0, 1, step - 1, step, step + 1, step + 2, 2 * step + 3, 3 * step + 1, 17, 5 * step, step, step + 5,
             step - 2, 0, 2 * step, 2 * step + 1, step + 7, 40001, step, 3, step + 9, 6 * step + 1, step, step, step + 1]
    # Every record is still (|d|^2 = 1) except a handful of voters: with VECTORS_NEEDED = 2 an "A" frame gives
    # two adjacent cells exactly 2 votes each (flag 1; one lost record -> 0), a "B" frame gives the second cell
    # a single vote (flag 0; one record counted twice -> 1).  Voters sit at the very first / last record (the
    # odd head / tail a lane handles alone), inside the first streaming step (the pre-issued one) and beyond it.
    frames, kinds = [], []
    for i, n in enumerate(sizes):
        mv = np.zeros(n, dtype=m.MV_DTYPE)
        mv["dst_x"] = rng.randint(0, w, size=n)
        mv["dst_y"] = rng.randint(0, h, size=n)
        mv["src_x"], mv["src_y"] = mv["dst_x"] - 1, mv["dst_y"]
        kind = "AB"[i % 2] if n >= 4 else "-"
        if kind != "-":
            cx, cy = int(rng.randint(2, w // 16 - 3)), int(rng.randint(h // 160 + 2, h // 16 - h // 160 - 3))
            pos = {0, n - 1, int(rng.randint(0, min(n, step))), int(rng.randint(n // 2, n))}
            while len(pos) < 4:
                pos.add(int(rng.randint(0, n)))
            pos = sorted(pos)
            cells = [(cx, cy), (cx + 1, cy), (cx, cy), (cx + 1, cy)]
            rng.shuffle(cells)
            if kind == "B":
                cells[cells.index((cx + 1, cy))] = (cx + 5, cy + 3)       # the second cell keeps ONE vote
            for q, (gx, gy) in zip(pos, cells):
                mv["dst_x"][q], mv["dst_y"][q] = gx * 16 + 8, gy * 16 + 8
                mv["src_x"][q], mv["src_y"][q] = gx * 16 + 8 - 6, gy * 16 + 8       # |d|^2 = 36 >= 16
        frames.append(mv if i % 9 != 4 else None)         # every 9th frame: no side data at all
        kinds.append(kind if i % 9 != 4 else "-")
    b = m.FrameBatch.from_frames(frames)
    assert len(set(int(x) & 1 for x in b.frame_off)) == 2                     # both alignments of a frame start occur
    rec = m.pack_records(b.mv)
    d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda()
    d_off = torch.from_numpy(b.frame_off.astype(np.int64)).cuda()
    d_sd = torch.from_numpy(b.has_sd).cuda()
    for kw in (dict(vectors_needed=2, clusters_needed=1), dict(vectors_needed=1, clusters_needed=2), dict(vectors_needed=0)):
        p = ob.params_from_config(w, h, **kw)
        want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
        want_nosd = ob.scan_frames(p, b.mv, b.frame_off, None)
        if kw["vectors_needed"] == 2:
            assert want.tolist() == [1 if k == "A" else 0 for k in kinds]
        for g, pf in ((2, 1), (3, 1), (4, 1), (8, 1), (25, 1), (64, 1), (4, 0), (1, 1)):
            if pf == 0 and not experiments_build():           # MTGPU_PREFETCH=0 / MTGPU_ALIGN=0: experiments build only
                continue
            monkeypatch.setenv("MTGPU_GROUP", str(g))
            monkeypatch.setenv("MTGPU_PREFETCH", str(pf))
            monkeypatch.setenv("MTGPU_ALIGN", str(pf))            # without the prefetch also without the line alignment
            s = gpu_scanner_factory(m.ScanParams.from_config(w, h, **kw))
            monkeypatch.delenv("MTGPU_GROUP")
            monkeypatch.delenv("MTGPU_PREFETCH")
            monkeypatch.delenv("MTGPU_ALIGN")
            assert s.plan["block_threads"] == block and s.plan["counter_bits"] == 32
            got = s.check_frames_device_compact(d_rec, d_off, d_sd).cpu().numpy()
            assert np.array_equal(got, want), (grid, kw, g, pf)
            got = s.check_frames_device_compact(d_rec, d_off, None).cpu().numpy()
            assert np.array_equal(got, want_nosd), (grid, kw, g, pf, "has_sd NULL")
            # a window that starts at an odd record and ends inside a group
            lo, hi = 2, len(sizes) - 3
            got = s.check_frames_device_compact(d_rec, d_off[lo:hi + 1].contiguous(), d_sd[lo:hi].contiguous()).cpu().numpy()
            assert np.array_equal(got, want[lo:hi]), (grid, kw, g, pf, "window")


def test_scratch_pool_stats_and_trim(gpu_scanner_factory):
    """the per-context scratch pool kept every block until mtgpu_destroy.  A banded plan's spill queue is
    4 bytes per record of the batch: it shows up in mtgpu_get_stats, mtgpu_trim hands it back, and the next scan
    simply maps scratch again (same flags)."""
    import torch
    p, s = _fine_shipped_env_scanner(gpu_scanner_factory)            # 960x540, 2 spill bands
    spec = synth.spec_4k_fine_dense(seed=3)
    spec.events = [synth.Event(1, 4, 400, 300, 4, 4, 7, 1)]
    mv, off, pts, sd = synth.gen_stream(spec, 6)
    want = ob.scan_frames(p, mv, off, sd)
    d_mv = torch.from_numpy(mv.view(np.uint8).copy()).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    before = s.stats()
    assert before["private_pool"] == 1 and before["hip_streams"] == 1
    got = s.check_frames_device(d_mv, d_off, None).cpu().numpy()
    assert np.array_equal(got, want) and 0 < want.sum() < 6
    torch.cuda.synchronize()
    held = s.stats()
    assert held["pool_reserved_high"] >= 4 * len(mv) and held["pool_reserved_bytes"] >= 4 * len(mv)     # the queue stays cached
    s.trim()
    after = s.stats()
    assert after["pool_reserved_bytes"] < held["pool_reserved_bytes"] and after["pool_reserved_high"] >= held["pool_reserved_high"]
    assert np.array_equal(s.check_frames_device(d_mv, d_off, None).cpu().numpy(), want)
    # the host-pointer entry points keep grow-only staging of their own: reported separately
    assert np.array_equal(s.check_frames(m.FrameBatch(mv, off, None, sd)), want)
    assert s.stats()["staging_device_bytes"] >= 40 * len(mv)


def test_trim_while_other_threads_scan(gpu_scanner_factory):
    """mtgpu_trim is "safe at any time, from any thread": two threads scan banded batches (each launch holds a scratch
    block: work list + spill queue) on their own streams while a third trims the context's scratch ring in a loop —
    blocks in use stay, idle ones go, launches that follow allocate again; every flag right, and the ring ends empty."""
    import threading
    import torch
    p, s = _fine_shipped_env_scanner(gpu_scanner_factory)            # 960x540, 2 spill bands
    spec = synth.spec_4k_fine_dense(seed=5)
    spec.events = [synth.Event(1, 5, 300, 200, 5, 4, 7, 1)]
    errors, lock, stop = [], threading.Lock(), threading.Event()

    def scan(w):
        try:
            mv, off, pts, sd = synth.gen_stream(spec, 4 + w, first_frame=w)
            want = ob.scan_frames(p, mv, off, sd)
            st = torch.cuda.Stream()
            d_mv = torch.from_numpy(mv.view(np.uint8).copy()).cuda()
            d_off = torch.from_numpy(off.astype(np.int64)).cuda()
            d_sd = torch.from_numpy(sd).cuda()
            torch.cuda.synchronize()
            for _ in range(25):
                got = s.check_frames_device(d_mv, d_off, d_sd, stream=st.cuda_stream)
                st.synchronize()
                assert np.array_equal(got.cpu().numpy(), want)
        except BaseException as e:          # noqa: BLE001 - reported to the main thread
            with lock:
                errors.append((w, repr(e)))

    def trim():
        while not stop.is_set():
            s.trim()

    threads = [threading.Thread(target=scan, args=(w,)) for w in range(2)]
    tt = threading.Thread(target=trim)
    tt.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    tt.join()
    assert not errors, errors
    torch.cuda.synchronize()
    s.trim()
    assert s.stats()["pool_reserved_bytes"] == 0


def test_plain_c_example(tmp_path):
    """examples/scan_example.c: the ABI consumed from plain C (gcc), end to end on the GPU."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(m.LIB_PATH)
    exe = str(tmp_path / "scan_example")
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "scan_example.c"), "-o", exe, "-L" + pkg, "-lmtgpu",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "motion frames 30, segments 1, do_cut 1" in out.stdout and "[0.500, 2.467]" in out.stdout


def test_cpp_decoder_adapter_example(tmp_path):
    """examples/decoder_adapter_example.cpp: a custom FrameSource (what a decoder adapter looks like) driving
    run_scan_pipeline.  The scene is rendered again here, in Python, and pushed through the oracle-driven
    transcription of the reference's worker loop: motion frames, segments and savings must agree bit for bit,
    with 1 and with 4 workers."""
    import os
    import subprocess
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(m.LIB_PATH)
    exe = str(tmp_path / "decoder_adapter_example")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(root, "include"),
                           "-I" + os.path.join(pkg, "csrc", "host"),
                           os.path.join(root, "examples", "decoder_adapter_example.cpp"), "-o", exe, "-L" + pkg,
                           "-lmtgpu", "-lpthread", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    W, H, FPS, GOP, N, TB = 1280, 720, 25, 25, 1500, 12800
    gw, gh = (W + 15) // 16, (H + 15) // 16

    def object_at(f):
        t = f / FPS
        if 8.0 <= t < 14.0:
            return 4 + int((t - 8.0) * 10.0), 20
        if 40.0 <= t < 43.0:
            return 60 - int((t - 40.0) * 12.0), 9
        return None

    ys, xs = np.mgrid[0:gh, 0:gw]
    frames = []
    for f in range(N):
        if f % GOP == 0:
            frames.append(None)
            continue
        mv = np.zeros(gw * gh, dtype=m.MV_DTYPE)
        mv["dst_x"], mv["dst_y"] = (xs * 16 + 8).ravel(), (ys * 16 + 8).ravel()
        dx = np.zeros((gh, gw), dtype=np.int64)
        dy = np.zeros((gh, gw), dtype=np.int64)
        o = object_at(f)
        if o:
            dx[o[1]:o[1] + 3, o[0]:o[0] + 4], dy[o[1]:o[1] + 3, o[0]:o[0] + 4] = 6, -1
        if f & 1:
            dx[30, 70], dy[30, 70] = -9, 4
        mv["src_x"], mv["src_y"] = mv["dst_x"] - dx.ravel(), mv["dst_y"] - dy.ravel()
        frames.append(mv)
    ticks = [f * (TB // FPS) for f in range(N)]
    spec = types.SimpleNamespace(fps=float(FPS), tb_den=TB)
    p = ob.params_from_config(W, H, vectors_needed=1)
    mp = m.MergeParams(duration=N / FPS)                       # code defaults: gap 5 s, padding 0.5 s, 5 %
    pooled, (want_seg, want_res) = _reference_worker_loop(spec, frames, ticks, N / FPS, p, 30.0, 0.0, mp)
    assert len(pooled) == 216 and len(want_seg) == 2 and want_res["do_cut"] == 1        # 144 + 72 frames, two passes
    env = {k: v for k, v in os.environ.items() if k not in ("VECTORS_NEEDED", "CHUNK_DURATION_SEC", "TARGET_FPS",
                                                            "MAX_GAP_SEC", "PADDING_SEC", "MIN_SAVINGS_PCT",
                                                            "MV_THRESHOLD_SQ", "CLUSTERS_NEEDED", "VERTICAL_MASK")}
    for threads in ("1", "4"):
        out = subprocess.run([exe, threads], check=True, capture_output=True, text=True, env=env).stdout.splitlines()
        head = dict(zip(out[0].split()[0::2], out[0].split()[1::2]))
        assert (head["chunks"], head["motion_frames"]) == ("2", "216") and int(head["frames_scanned"]) == N
        res = dict(zip(out[1].split()[0::2], out[1].split()[1::2]))
        assert int(res["do_cut"]) == 1
        assert float(res["time_removed"]).hex() == float(want_res["time_removed"]).hex()
        assert float(res["saved_pct"]).hex() == float(want_res["saved_pct"]).hex()
        segs = [[float(x).hex() for x in ln.split()[1:]] for ln in out[2:]]
        assert segs == [[float(a).hex(), float(b).hex()] for a, b in want_seg.tolist()]


def test_plain_c_pipe_example(tmp_path):
    """examples/pipe_example.c: the decoder-thread usage of the pinned pipe (acquire / add_frame /
    submit / collect / release with back-pressure) from plain C, end to end on the GPU."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(m.LIB_PATH)
    exe = str(tmp_path / "pipe_example")
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "pipe_example.c"), "-o", exe, "-L" + pkg, "-lmtgpu",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "motion frames 58, segments 2, do_cut 1" in out.stdout
    assert "[1.533, 3.467]" in out.stdout and "[7.533, 9.467]" in out.stdout


def test_one_context_entered_from_many_threads(gpu_scanner_factory):
    """include/mtgpu.h promises that every entry point may be called concurrently on ONE context (the
    reference enters its scanner code from N x S host threads, motion_scanner.hpp:8-13): eight threads
    hammer the same context with the synchronous scan, the device-resident scan on their own streams
    (40-byte and compact), a pipe each, and merges (small and multi-workgroup), all checked against the oracle."""
    import threading
    import torch
    spec = synth.spec_1080p(seed=23, sub=1)
    spec.events = synth.scripted_events(spec, 90)
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    mp = m.MergeParams(400.0, 1.0, 0.25, 5.0)
    errors, lock = [], threading.Lock()

    def worker(w):
        try:
            rng = np.random.default_rng(100 + w)
            frames = [synth.gen_frame(spec, int(i)) for i in rng.integers(0, 90, size=40)]
            b = m.FrameBatch.from_frames(frames)
            want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
            st = torch.cuda.Stream()
            d_mv = torch.from_numpy(b.mv.view(np.uint8).reshape(-1).copy()).cuda()
            d_rec = torch.from_numpy(m.pack_records(b.mv).view(np.uint8).reshape(-1).copy()).cuda()
            d_off = torch.from_numpy(b.frame_off.astype(np.int64)).cuda()
            d_sd = torch.from_numpy(b.has_sd).cuda()
            torch.cuda.synchronize()
            ts = np.sort(rng.uniform(0, 400, size=300 if w % 2 else 9000))
            seg_want, res_want = ob.pool_and_merge(ts, mp, True)
            for it in range(6):
                assert np.array_equal(s.check_frames(b), want)
                f1 = s.check_frames_device(d_mv, d_off, d_sd, stream=st.cuda_stream)
                f2 = s.check_frames_device_compact(d_rec, d_off, d_sd, stream=st.cuda_stream)
                pipe = m.ScanPipe(s, 8160 * 6, 16, 2, layout=(m.LAYOUT_COMPACT8 | m.LAYOUT_ZERO_COPY) if it % 2 else m.LAYOUT_AOS40)
                for i, f in enumerate(frames):
                    pipe.feed(f, float(i), tag=i)
                out = pipe.drain()
                pipe.close()
                assert [fl for _, fl, _ in out] == want.tolist()
                seg, res = s.merge_segments(rng.permutation(ts), mp, job_semantics=True)
                assert np.array_equal(seg.view(np.uint64), seg_want.view(np.uint64)) and res["n_segments"] == res_want["n_segments"]
                st.synchronize()
                assert np.array_equal(f1.cpu().numpy(), want) and np.array_equal(f2.cpu().numpy(), want)
        except BaseException as e:          # noqa: BLE001 - reported to the main thread
            with lock:
                errors.append((w, repr(e)))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_settings_change_while_other_threads_scan(gpu_scanner_factory):
    """mtgpu_set_slices and mtgpu_profile_enable may be called on a context that other threads are scanning through
    (the host layer shares one context per (device, parameters) across S x T workers; the reference enters its
    scanner from N workers at once, src/pipeline.cpp:186-197): four threads scan device-resident batches — few large
    frames, so that the slices setting really changes the launch — while a fifth keeps changing both settings;
    every flag equals the oracle's, every time."""
    import threading
    import torch
    spec = synth.spec_4k(seed=31, sub=2)
    spec.events = synth.scripted_events(spec, 40)
    p = ob.params_from_config(3840, 2160)
    s = gpu_scanner_factory(p)
    errors, lock, stop = [], threading.Lock(), threading.Event()

    def scan(w):
        try:
            frames = [synth.gen_frame(spec, 30 + (w * 3 + i) % 10) for i in range(6)]
            b = m.FrameBatch.from_frames(frames)
            want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
            st = torch.cuda.Stream()
            d_mv = torch.from_numpy(b.mv.view(np.uint8).reshape(-1).copy()).cuda()
            d_off = torch.from_numpy(b.frame_off.astype(np.int64)).cuda()
            d_sd = torch.from_numpy(b.has_sd).cuda()
            torch.cuda.synchronize()
            for _ in range(40):
                got = s.check_frames_device(d_mv, d_off, d_sd, stream=st.cuda_stream)
                st.synchronize()
                assert np.array_equal(got.cpu().numpy(), want)
        except BaseException as e:          # noqa: BLE001 - reported to the main thread
            with lock:
                errors.append((w, repr(e)))

    def toggle():
        k = 0
        while not stop.is_set():
            s.set_slices((0, 1, 2, 4, 8)[k % 5])
            s.profile(k % 3 == 0)
            if k % 7 == 0:
                s.profile_read()
            k += 1

    threads = [threading.Thread(target=scan, args=(w,)) for w in range(4)]
    tg = threading.Thread(target=toggle)
    tg.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    tg.join()
    s.set_slices(0)
    s.profile(False)
    s.profile_read()
    assert not errors, errors


def test_many_threads_and_streams_never_share_launch_scratch(gpu_scanner_factory):
    """Every launch of the device entry points takes a scratch block (its work list) from the context's ring; sixteen
    threads, each on its own stream, launch small batches back to back through ONE context — three different batches per
    thread in turn, so that a launch that read another launch's list would answer for the wrong frames.  (The
    runtime's stream-ordered allocator, which served this scratch through round 5, did exactly that under this load:
    profiles/r06_host_stress_*.log.)  2400 launches, every flag compared with the oracle."""
    import threading
    import torch
    spec = synth.spec_1080p(seed=41, sub=1)
    spec.events = synth.scripted_events(spec, 90)
    p = ob.params_from_config(1920, 1080, vectors_needed=1)
    s = gpu_scanner_factory(p)
    errors, lock = [], threading.Lock()

    def worker(w):
        try:
            rng = np.random.default_rng(500 + w)
            st = torch.cuda.Stream()
            batches = []
            for _ in range(3):
                frames = [synth.gen_frame(spec, int(i)) for i in rng.integers(0, 90, size=int(rng.integers(3, 17)))]
                b = m.FrameBatch.from_frames(frames)
                batches.append((torch.from_numpy(b.mv.view(np.uint8).reshape(-1).copy()).cuda(),
                                torch.from_numpy(b.frame_off.astype(np.int64)).cuda(), torch.from_numpy(b.has_sd).cuda(),
                                ob.scan_frames(p, b.mv, b.frame_off, b.has_sd),
                                torch.empty(len(frames), dtype=torch.uint8, device="cuda")))
            torch.cuda.synchronize()
            for it in range(50):
                for d_mv, d_off, d_sd, want, d_flags in batches:          # three launches in flight per stream
                    s.check_frames_device(d_mv, d_off, d_sd, d_flags, stream=st.cuda_stream)
                st.synchronize()
                for d_mv, d_off, d_sd, want, d_flags in batches:
                    assert np.array_equal(d_flags.cpu().numpy(), want), (w, it)
        except BaseException as e:          # noqa: BLE001 - reported to the main thread
            with lock:
                errors.append((w, repr(e)))

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    assert s.stats()["pool_reserved_bytes"] > 0


def test_device_entry_points_under_real_thread_concurrency(tmp_path):
    """tests/cpp/device_entry_stress.cpp: sixteen std::threads (no interpreter lock between them), each on its own HIP
    stream, 150 x 3 launches of small batches through ONE context — the launches' work lists come from the context's
    scratch ring.  Expected flags are known by construction; every launch is compared.  (This is the load under which
    the runtime's stream-ordered allocator let launches share scratch: profiles/r06_host_stress_*.log.)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(m.LIB_PATH)
    exe = str(tmp_path / "device_entry_stress")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
                           os.path.join(root, "tests", "cpp", "device_entry_stress.cpp"), "-o", exe, "-L" + pkg, "-lmtgpu",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, "16", os.environ.get("MTGPU_STRESS_ITERS", "150")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr[-2000:]


def test_launch_timing_events(gpu_scanner_factory):
    """mtgpu_profile_enable / mtgpu_profile_read: one event triple per scan launch (planning | scan kernel), a ring of
    64 that a 65th launch drains by itself, totals reset by every read, nothing recorded while off."""
    import torch
    spec = synth.spec_1080p(seed=3, sub=2)
    spec.events = synth.scripted_events(spec, 40)
    mv, off, pts, sd = synth.gen_stream(spec, 40)
    p = ob.params_from_config(1920, 1080)
    s = gpu_scanner_factory(p)
    want = ob.scan_frames(p, mv, off, sd)
    d_mv = torch.from_numpy(mv.view(np.uint8).copy()).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    d_sd = torch.from_numpy(sd).cuda()
    s.check_frames_device(d_mv, d_off, d_sd)
    assert s.profile_read()["launches"] == 0                  # off: nothing recorded
    s.profile(True)
    for n in (5, 70):                                         # 70 > the ring of 64
        for _ in range(n):
            got = s.check_frames_device(d_mv, d_off, d_sd)
        pr = s.profile_read()
        assert pr["launches"] == n and 0.0 < pr["plan_ms"] < pr["scan_ms"] < 50.0, pr
        assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(s.check_frames(m.FrameBatch(mv, off, None, sd)), want)      # the host-pointer entry is timed too
    assert s.profile_read()["launches"] == 1
    s.profile(False)
    s.check_frames_device(d_mv, d_off, d_sd)
    assert s.profile_read() == {"plan_ms": 0.0, "scan_ms": 0.0, "launches": 0}


@pytest.mark.parametrize("job", [False, True])
def test_merge_extreme_finite_values(gpu_scanner_factory, job):
    """Timestamps and merge constants at the edges of binary64 (denormals, 1e308, negative times, a
    lone -0.0, infinite MAX_GAP / PADDING / duration): every IEEE operation of pipeline.cpp:323-356 must
    round the same way on the GPU (no FMA contraction, no flush-to-zero, libstdc++ max/min operand order)."""
    s = gpu_scanner_factory(ob.params_from_config(1920, 1080))
    inf = float("inf")
    lists = [
        [5e-324, 1e-323, 2.2250738585072014e-308, 1.0, 1.0000000000000002, 2.0],
        [-3.5, -0.0, 1.25, 7.0, 7.0 + 2 ** -50, 1e3],
        [0.1 * k for k in range(60)] + [1e308, 1.7976931348623157e308],
        [-1e308, -1.0, 0.0, 1e-300, 1e300],
        [2.0 ** 52 + k for k in range(8)] + [2.0 ** 53, 2.0 ** 53 + 2],
        [123.456],
    ]
    params = [
        m.MergeParams(duration=100.0, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0),
        m.MergeParams(duration=1e308, max_gap_sec=1e-320, padding_sec=5e-324, min_savings_pct=0.0),
        m.MergeParams(duration=inf, max_gap_sec=1.0, padding_sec=0.25, min_savings_pct=5.0),
        m.MergeParams(duration=50.0, max_gap_sec=inf, padding_sec=1.0, min_savings_pct=99.0),
        m.MergeParams(duration=50.0, max_gap_sec=0.0, padding_sec=inf, min_savings_pct=-1.0),
        m.MergeParams(duration=5e-324, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0),
        m.MergeParams(duration=-10.0, max_gap_sec=5.0, padding_sec=0.5, min_savings_pct=5.0),
    ]
    rng = np.random.RandomState(3)
    for ts in lists:
        for mp in params:
            merge_case(s, rng.permutation(np.array(ts, dtype=np.float64)), mp, job)


def test_device_offsets_precondition_check_opt_in(gpu_scanner_factory, monkeypatch):
    """include/mtgpu.h: d_frame_off must be non-decreasing (frames = disjoint record ranges; a banded plan keeps
    each frame's vote queue at the frame's own offset).  With MTGPU_CHECK_OFFSETS=1 (read at mtgpu_create) both
    device entry points verify that on the device and return MT_ERR_INVALID naming the first offending frame,
    before anything is scanned; well-formed offsets scan as usual — on the automatic 2-band plan of the 960x540
    grid (shipped env) and on a single-tile plan; without the knob nothing is checked (asynchronous API)."""
    import torch
    monkeypatch.setenv("MTGPU_CHECK_OFFSETS", "1")
    p, s = _fine_shipped_env_scanner(gpu_scanner_factory)
    spec = synth.spec_4k_fine_dense(seed=5)
    spec.events = [synth.Event(1, 3, 300, 100, 5, 4, 9, 3), synth.Event(3, 5, 640, 400, 4, 3, -7, 2)]
    mv, off, pts, sd = synth.gen_stream(spec, 6)
    want = ob.scan_frames(p, mv, off, sd)
    d_mv = torch.from_numpy(mv.view(np.uint8).copy()).cuda()
    d_sd = torch.from_numpy(sd.astype(np.uint8)).cuda()
    good = torch.from_numpy(off.astype(np.int64)).cuda()
    got = s.check_frames_device(d_mv, good, d_sd)
    assert np.array_equal(got.cpu().numpy(), want) and want.sum() >= 2
    bad = off.astype(np.int64).copy()
    bad[3] = bad[2] - 1000                                    # frame 2 would end before it starts; frame 3 overlaps frame 1..2
    flags = torch.full((6,), 7, dtype=torch.uint8, device="cuda")
    with pytest.raises(m.MtgpuError) as ei:
        s.check_frames_device(d_mv, torch.from_numpy(bad).cuda(), d_sd, flags=flags)
    assert ei.value.code == m._abi.MT_ERR_INVALID and "frame_off[2] > frame_off[3]" in str(ei.value)
    torch.cuda.synchronize()
    assert flags.cpu().tolist() == [7] * 6                    # nothing was scanned
    # the compact entry point checks too
    rec8 = torch.from_numpy(m.pack_records(mv).view(np.uint8).copy()).cuda()
    assert np.array_equal(s.check_frames_device_compact(rec8, good, d_sd).cpu().numpy(), want)
    with pytest.raises(m.MtgpuError) as ei:
        s.check_frames_device_compact(rec8, torch.from_numpy(bad).cuda(), d_sd)
    assert ei.value.code == m._abi.MT_ERR_INVALID
    # several bad entries: the FIRST one is named; single-tile plan
    p2 = ob.params_from_config(1920, 1080)
    s2 = gpu_scanner_factory(p2)
    rng = np.random.RandomState(3)
    mv2, off2, _ = synth.random_frames(rng, 3000, 300, 1920, 1080)
    bad2 = off2.astype(np.int64).copy()
    bad2[1500], bad2[2500] = bad2[1499] - 1, 0
    with pytest.raises(m.MtgpuError) as ei:
        s2.check_frames_device(torch.from_numpy(mv2.view(np.uint8).copy()).cuda(), torch.from_numpy(bad2).cuda())
    assert "frame_off[1499] > frame_off[1500]" in str(ei.value)
    # a context created without the knob does not check (and a single-tile plan tolerates the overlap: clamped ranges)
    monkeypatch.delenv("MTGPU_CHECK_OFFSETS")
    s3 = gpu_scanner_factory(p2)
    got3 = s3.check_frames_device(torch.from_numpy(mv2.view(np.uint8).copy()).cuda(), torch.from_numpy(bad2).cuda())
    torch.cuda.synchronize()
    assert got3.numel() == 3000


from run_frames import run_frames as _run_frames  # noqa: E402  (shared with the CPU-tier oracle cross-check)


@pytest.mark.parametrize("force_fb,vec", [(2, 1), (2, 2), (4, 3), (4, 4), (8, 5), (8, 8), (108, 9), (108, 40), (32, 4)])
def test_same_cell_runs_every_length_and_alignment(gpu_scanner_factory, force_fb, vec):
    """Round 4: runs of records that one wave instruction maps to the same cell vote once, with their count
    (scan_kernels.hip: vote / bump_n).  Runs of every length from 1 to beyond a wave at every lane alignment, across
    wave-instruction and streaming-step boundaries, in the divergent head / tail code, with cells ending exactly one
    vote short of / at / above VECTORS_NEEDED, for every packed counter form (2 / 4 / 8-bit thermometer, 8-bit CAS;
    where a field counts beyond 4 the runs are cut every 4 lanes) and the 32-bit form as control — against the
    oracle, as 40-byte and as compact records."""
    import torch
    p = ob.params_from_config(1920, 1080, vectors_needed=vec, clusters_needed=2, mv_threshold_sq=4.0)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    rng = np.random.RandomState(1000 * force_fb + vec)
    frames = _run_frames(rng, 1920, 1080, 4, vec, 96, 3)
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert 5 < want.sum() < len(want) - 5
    rec8 = torch.from_numpy(m.pack_records(b.mv).view(np.uint8).copy()).cuda()
    got8 = s.check_frames_device_compact(rec8, torch.from_numpy(b.frame_off.astype(np.int64)).cuda(),
                                         torch.from_numpy(b.has_sd.astype(np.uint8)).cuda())
    assert np.array_equal(got8.cpu().numpy(), want)


def test_same_cell_runs_across_the_band_seam(gpu_scanner_factory):
    """The same on the automatic 2-band plan of the 960x540 grid (4-bit fields, VECTORS_NEEDED 4): a queue entry
    now stands for a run of up to 4 records — cells in the seam rows 268..271 (band 1 sees their votes only through
    replayed entries) collect 3 / 4 / 5 / many votes in runs of 1..130 records; every flag as the oracle says."""
    p, s = _fine_shipped_env_scanner(gpu_scanner_factory)
    rng = np.random.RandomState(77)
    frames = []
    for trial in range(64):
        runs = []
        for _ in range(int(rng.randint(1, 3))):
            gx = int(rng.randint(1, 957))
            gy = int(rng.choice([268, 269, 270, 300, 100]))
            for (cx, cy) in ((gx, gy), (gx, gy + 1)):
                left = int(rng.choice([3, 4, 5, 84, 700]))
                while left > 0:
                    n = min(left, int(rng.choice([1, 1, 2, 3, 4, 5, 8, 9, 64, 65, 130])))
                    runs.append((cx, cy, n))
                    left -= n
        recs = []
        for i in rng.permutation(len(runs)):
            cx, cy, n = runs[i]
            k = int(rng.choice([0, 1, 3, 64, 700]))
            recs += [(int(a), int(b), 1) for a, b in zip(rng.randint(0, 3840, size=k), rng.randint(0, 2160, size=k))]
            recs += [((cx << 2) + int(rng.randint(0, 4)), (cy << 2) + int(rng.randint(0, 4)), 3) for _ in range(n)]
        arr = np.array(recs, dtype=np.int64).reshape(-1, 3)
        mv = np.zeros(len(arr), dtype=m.MV_DTYPE)
        mv["dst_x"], mv["dst_y"] = arr[:, 0], arr[:, 1]
        mv["src_x"], mv["src_y"] = arr[:, 0] - arr[:, 2], arr[:, 1]
        frames.append(mv)
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert 5 < want.sum() < len(want) - 5


def test_throwing_decoder_fails_its_video_and_returns_the_cpu_token(tmp_path):
    """Round 4 (CpuGate): an exception out of FrameSource::next() reaches the worker while it holds a half-filled
    staging batch and a CPU token.  The video must fail with the decoder's message, and the token must come back —
    with MTGPU_CPU_TOKENS=1 the other two workers would otherwise wait for it for ever (the test would time out).
    The same scene without the fault scans to the end."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(m.LIB_PATH)
    exe = str(tmp_path / "throwing_source")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(root, "include"),
                           "-I" + os.path.join(pkg, "csrc", "host"), os.path.join(root, "tests", "cpp", "throwing_source.cpp"),
                           "-o", exe, "-L" + pkg, "-lmtgpu", "-lpthread", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    env = dict(os.environ, MTGPU_CPU_TOKENS="1", CHUNK_DURATION_SEC="10", TARGET_FPS="0", VECTORS_NEEDED="1")
    out = subprocess.run([exe, "3", "777"], env=env, capture_output=True, text=True, timeout=120).stdout
    assert out.startswith("rc 1 ") and "tokens 1 " in out and "decoder lost the stream at frame 777" in out, out
    out = subprocess.run([exe, "3", "-1"], env=env, capture_output=True, text=True, timeout=120).stdout
    assert out.startswith("rc 0 motion 700 tokens 1 "), out          # frames 100..199, 300..399, ...: 7 x 100 moving frames


def test_recorded_wrong_flag_configuration_through_every_path(gpu_scanner_factory, monkeypatch):
    """The one configuration in which this library ever returned a wrong answer (round 4, with staging that was
    page-locked by hipHostRegister: profiles/r04_soak_mismatch_with_registered_staging.txt, reconstructed by
    tests/soak_replay.py) through all three paths — 40-byte records, device-resident compact records (the soak had NOT
    run that path in the failing iteration, so the REC 8 / 8-bit CAS / two-frames-per-workgroup kernel was only ever
    seen through the pipe), and the zero-copy pipe with the recorded geometry, a fresh pipe each time so that frame 33
    again sits in the first use of a freshly pinned block."""
    import torch
    from soak_replay import replay
    head, tail, plan, p = replay(10242, 70, pipe_every=5)
    mv, off, sd = tail["mv"], tail["off"], tail["sd"]
    want = ob.scan_frames(p, mv, off, sd)
    assert want[33] == 1 and want[32] == 1
    for k_, v_ in head["knobs"].items():
        if v_:
            monkeypatch.setenv(k_, v_)
    s = gpu_scanner_factory(p, force_fb=head["force_fb"])
    for k_ in head["knobs"]:
        monkeypatch.delenv(k_, raising=False)
    s.set_slices(tail["slices"])
    assert s.plan["counter_mode"] == 2 and s.plan["counter_bits"] == 8
    rec = m.pack_records(mv)
    d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    d_sd = torch.from_numpy(sd).cuda()
    for rep in range(3):
        assert np.array_equal(s.check_frames(m.FrameBatch(mv, off, None, sd)), want), rep
        assert np.array_equal(s.check_frames_device_compact(d_rec, d_off, d_sd).cpu().numpy(), want), ("compact", rep)
        # the second half alone, as the pipe's second batch shipped it (frame 33 = item 1 of the launch)
        got = s.check_frames_device_compact(d_rec, d_off[32:].contiguous(), d_sd[32:].contiguous()).cpu().numpy()
        assert np.array_equal(got, want[32:]), ("compact, frames 32..63", rep)
        pipe = m.ScanPipe(s, *tail["pipe"])
        for f in range(tail["n_frames"]):
            fr = mv[int(off[f]):int(off[f + 1])]
            pipe.feed(fr if sd[f] else None, float(f), tag=f)
        out = pipe.drain()
        pipe.close()
        assert [fl for _, fl, _ in out] == want.tolist(), ("pipe", rep)


def _raster_frame(rows, gw, mult, special=(), drop=(), lead=0, shift=2, step_px=3):
    """Records in raster order over grid rows `rows`, `mult` records per cell (all above the threshold), plus extra
    records for the cells in `special` ((gx, gy) -> extra votes, appended right behind the cell's own), minus the
    cells in `drop`; `lead` non-voting records in front (shifts every wave instruction against the rows)."""
    recs = [(1, 1, 0)] * lead
    for gy in rows:
        for gx in range(gw):
            if (gx, gy) in drop:
                continue
            n = mult + dict(special).get((gx, gy), 0)
            recs += [((gx << shift) + 1, (gy << shift) + 1, step_px)] * n
    a = np.array(recs, dtype=np.int64).reshape(-1, 3)
    mv = np.zeros(len(a), dtype=m.MV_DTYPE)
    mv["dst_x"], mv["dst_y"] = a[:, 0], a[:, 1]
    mv["src_x"], mv["src_y"] = a[:, 0] - a[:, 2], a[:, 1]
    return mv


def test_dense_raster_votes_travel_as_spans_and_count_exactly(gpu_scanner_factory):
    """Round 5: on banded plans a wave instruction whose queued votes fall into consecutive cells of a row is ONE
    8-byte span entry (scan_kernels.hip, SpillQ), replayed a counter word at a time.  960x540 / VECTORS_NEEDED 4 /
    2 bands: whole rows on both sides of the band seam (rows 262..279; band 1 starts at 270) vote with 3 records per
    cell — one short of active — in raster order; two horizontally adjacent cells get a 4th record, which makes
    exactly two active neighbours = the 2 clusters the frame needs.  One vote too many or too few anywhere in the
    replay flips the flag: frames WITHOUT the pair must say no (3 votes everywhere), frames WITH it yes, wherever the
    pair sits (word boundaries of the 4-bit fields, row ends, the seam rows), however the wave instructions are
    shifted against the rows (0..70 leading records, 1..5 records per cell), with holes that cut spans short."""
    import torch
    p, s = _fine_shipped_env_scanner(gpu_scanner_factory)
    gw = 960
    rows = list(range(262, 280))
    frames, expect = [], []
    spots = [(5, 275), (7, 275), (8, 275), (15, 275), (16, 275), (63, 276), (64, 276), (957, 277), (956, 277), (2, 278),
             (1, 278), (400, 269), (401, 270), (402, 271), (333, 268), (500, 262), (501, 279)]
    for i, (gx, gy) in enumerate(spots):
        lead = [0, 1, 2, 3, 5, 13, 17, 31, 63, 64, 65, 70][i % 12]
        pair = (((gx, gy), 1), ((gx + 1, gy), 1)) if gx + 1 < gw else (((gx - 1, gy), 1), ((gx, gy), 1))
        holes = {(int(x), int(y)) for x, y in zip(np.random.RandomState(i).randint(0, gw, 40), np.random.RandomState(i + 99).choice(rows, 40))}
        holes -= {c for c, _ in pair}
        frames.append(_raster_frame(rows, gw, 3, special=pair, drop=holes if i % 3 == 0 else (), lead=lead))
        expect.append(1)
        frames.append(_raster_frame(rows, gw, 3, drop=holes if i % 3 == 0 else (), lead=lead))          # no pair: 3 votes everywhere
        expect.append(0)
    for mult, lead in ((1, 7), (2, 9), (4, 3), (5, 11)):                   # every cell 1 / 2 records: no; 4 / 5 records: every cell active
        frames.append(_raster_frame(rows, gw, mult, lead=lead))
        expect.append(1 if mult >= 4 else 0)
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert want.tolist() == expect
    rec8 = torch.from_numpy(m.pack_records(b.mv).view(np.uint8).copy()).cuda()
    got8 = s.check_frames_device_compact(rec8, torch.from_numpy(b.frame_off.astype(np.int64)).cuda(), None)
    assert got8.cpu().numpy().tolist() == expect


@pytest.mark.parametrize("vec,force_fb", [(3, None), (6, None), (8, None), (12, None), (4, 32)])
def test_spans_on_other_banded_counter_forms(gpu_scanner_factory, vec, force_fb):
    """The same dense raster input on the other forms a banded plan can take: 4-bit fields at VECTORS_NEEDED 3, 8-bit
    thermometer fields (runs are cut every 4 lanes there: equal cells follow each other, no spans), 8-bit CAS fields
    (VECTORS_NEEDED 12) and forced 32-bit counters (many narrow bands) — the span replay's generic cell-by-cell path."""
    kw = dict(m.config.SHIPPED_ENV)
    kw.update(block_size=4, block_shift=2, vectors_needed=vec)
    p = ob.params_from_config(3840, 2160, **kw)
    s = gpu_scanner_factory(p, force_fb=force_fb)
    assert s.plan["bands"] >= 2
    band_rows = s.plan["band_rows"]
    seam = p.vertical_margin + band_rows                      # first centre row of band 1
    rows = list(range(seam - 4, seam + 5))
    frames, expect = [], []
    for i, gx in enumerate((3, 8, 62, 500, 957)):
        gy = rows[(2 * i) % len(rows)]
        pair = (((gx, gy), 1), ((gx + 1, gy), 1))
        frames.append(_raster_frame(rows, 960, vec - 1, special=pair, lead=5 * i + 1))
        expect.append(1)
        frames.append(_raster_frame(rows, 960, vec - 1, lead=5 * i + 1))
        expect.append(0)
    b = m.FrameBatch.from_frames(frames)
    want = assert_scan_parity(s, p, b.mv, b.frame_off, b.has_sd)
    assert want.tolist() == expect


def test_device_entry_points_with_flags_in_pinned_host_memory(gpu_scanner_factory):
    """include/mtgpu.h, "Memory the `*_device` entry points accept": result bytes may live in pinned host memory that
    the driver allocated (torch's pin_memory = hipHostMalloc); the entry points ask the runtime what `d_flags` is and
    store at system scope then.  Records resident on the device, flags written over PCIe into a pinned buffer with its
    own 128-byte lines, read by the host after a stream synchronisation — for single-tile, sliced and banded plans and
    for both record layouts; a second call into the same pinned buffer overwrites every byte."""
    import torch
    spec = synth.spec_1080p(seed=31, sub=2)
    spec.events = synth.scripted_events(spec, 90)
    mv, off, pts, sd = synth.gen_stream(spec, 90)
    cases = [(ob.params_from_config(1920, 1080), None, 0), (ob.params_from_config(1920, 1080, vectors_needed=2), 2, 4)]
    kw = dict(m.config.SHIPPED_ENV)
    kw.update(block_size=4, block_shift=2)
    cases.append((ob.params_from_config(3840, 2160, **kw), None, 0))          # 2 spill bands
    d_mv = torch.from_numpy(mv.view(np.uint8).reshape(-1).copy()).cuda()
    d_rec = torch.from_numpy(m.pack_records(mv).view(np.uint8).reshape(-1).copy()).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    d_sd = torch.from_numpy(sd).cuda()
    seen = []
    for p, fb, slices in cases:
        s = gpu_scanner_factory(p, force_fb=fb)
        s.set_slices(slices)
        want = ob.scan_frames(p, mv, off, sd)
        seen.append(int(want.sum()))
        host = torch.empty(256, dtype=torch.uint8).pin_memory()
        assert host.data_ptr() % 128 == 0
        flags = host[:90]
        for fill in (7, 9):
            host.fill_(fill)
            s.check_frames_device(d_mv, d_off, d_sd, flags)
            torch.cuda.synchronize()
            assert np.array_equal(flags.numpy(), want) and int(host[90]) == fill, (s.plan, "40-byte records")
            host.fill_(fill)
            s.check_frames_device_compact(d_rec, d_off, d_sd, flags)
            torch.cuda.synchronize()
            assert np.array_equal(flags.numpy(), want) and int(host[90]) == fill, (s.plan, "compact records")
    assert 0 < seen[0] < 90 and 0 < seen[1] < 90          # (the 1080p stream on the 960x540 grid of the third case never reaches 4 votes)


@pytest.mark.parametrize("slices,group,chunk", [(0, "", ""), (2, "", ""), (0, "3", ""), (0, "", "5"), (0, "2", "7")])
@pytest.mark.parametrize("period", [2, 8, 30])
def test_work_list_skips_frames_without_side_data(gpu_scanner_factory, monkeypatch, slices, group, chunk, period):
    """The work list (plan_count_kernel / plan_scatter_kernel): frames WITHOUT side data are answered by the planning
    kernels (false, src/motion_scanner.cpp:219-221) and never get a workgroup; frames with side data — also with EMPTY
    side data — keep their stream order in the list.  Key frames every 2 / 8 / 30 frames (period 8 used to leave one XCD
    without work), a batch that is not a multiple of 8, frame slices, several frames per workgroup, launches cut into
    chunks of 5 and 7 workgroups, 40-byte and compact records, has_sd given and has_sd == NULL."""
    import torch
    for k_, v_ in (("MTGPU_GROUP", group), ("MTGPU_ITEM_CHUNK", chunk)):
        if v_:
            monkeypatch.setenv(k_, v_)
    p = ob.params_from_config(1920, 1080, vectors_needed=2, clusters_needed=1)
    s = gpu_scanner_factory(p)
    for k_ in ("MTGPU_GROUP", "MTGPU_ITEM_CHUNK"):
        monkeypatch.delenv(k_, raising=False)
    s.set_slices(slices)
    rng = np.random.RandomState(11 + period)
    frames = []
    for f in range(203):                                    # 203 = 25 octets + 3
        if f % period == 0:
            frames.append(None if f % (2 * period) == 0 else np.zeros(0, dtype=m.MV_DTYPE))
            continue
        n = int(rng.choice([1, 40, 700, 3000]))
        mv = np.zeros(n, dtype=m.MV_DTYPE)
        mv["dst_x"], mv["dst_y"] = rng.randint(0, 1920, size=n), rng.randint(0, 1080, size=n)
        mv["src_x"], mv["src_y"] = mv["dst_x"] - rng.choice([0, 1, 5], size=n), mv["dst_y"]
        if f % 3 == 0:                                      # a cluster: two votes in each of two neighbouring cells
            for q, (gx, gy) in enumerate(((50, 30), (50, 30), (51, 30), (51, 30))):
                if q < n:
                    mv["dst_x"][q], mv["dst_y"][q], mv["src_x"][q], mv["src_y"][q] = gx * 16 + 8, gy * 16 + 8, gx * 16, gy * 16 + 8
        frames.append(mv)
    b = m.FrameBatch.from_frames(frames)
    want = ob.scan_frames(p, b.mv, b.frame_off, b.has_sd)
    assert 10 < want.sum() < 150
    rec8 = torch.from_numpy(m.pack_records(b.mv).view(np.uint8).copy()).cuda()
    d_off = torch.from_numpy(b.frame_off.astype(np.int64)).cuda()
    d_sd = torch.from_numpy(b.has_sd.astype(np.uint8)).cuda()
    for _ in range(2):                                      # twice: the scratch of the first launch is reused
        assert np.array_equal(s.check_frames(b), want)
        assert np.array_equal(s.check_frames_device_compact(rec8, d_off, d_sd).cpu().numpy(), want), "compact"
    # has_sd == NULL: side data iff records (with vectors_needed >= 1 a frame with empty side data is false either way)
    assert np.array_equal(s.check_frames_device_compact(rec8, d_off, None).cpu().numpy(), want), "compact, has_sd NULL"
    # every flag is WRITTEN, also the ones the planning kernels answer: start from a poisoned buffer
    d_flags = torch.full((203,), 7, dtype=torch.uint8, device="cuda")
    s.check_frames_device_compact(rec8, d_off, d_sd, d_flags)
    assert np.array_equal(d_flags.cpu().numpy(), want)
