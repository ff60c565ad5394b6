"""Host-side mirror of the reference's scanner interface for the MV-scan path.

`MotionScanner` keeps the public shape of the reference class
(include/motion_trim/motion_scanner.hpp:113-152: construct, initialize(), scan_range())
but its `check_frame` half runs on the MI355X through the C ABI of include/mtgpu.h.
FFmpeg decode stays on the host and is NOT part of this package: frames arrive as
already-extracted AVMotionVector arrays (`FrameBatch`).

Host logic restated here (cheap, sequential, per call):
  - cfg/grid derivation   -> done inside libmtgpu (mtgpu_params_from_config)
  - scan_range's frame filter (src/motion_scanner.cpp:307-314, 357-371)
  - chunk creation (src/pipeline.cpp:141-142, 163-167)
Everything per-MV or per-timestamp runs in HIP kernels.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi, config
from ._abi import (COMPACT_DTYPE, LAYOUT_AOS40, LAYOUT_COMPACT8, LAYOUT_ZERO_COPY, MERGE_PARAMS_DTYPE, MERGE_RESULT_DTYPE,
                   MV_DTYPE, SEGMENT_DTYPE, MergeParamsC, MergeResultC, PlanC, ScanParamsC, check,
                   load_library)


@dataclass
class ScanParams:
    """mt_scan_params (include/mt_types.h) == MotionScanner::cfg + grid dims."""
    mv_threshold_sq: float
    block_shift: int
    clusters_needed: int
    vertical_margin: int
    vectors_needed: int
    grid_w: int
    grid_h: int

    @classmethod
    def from_config(cls, width: int, height: int, mv_threshold_sq=None, block_size=None,
                    block_shift=None, vectors_needed=None, clusters_needed=None,
                    vertical_mask=None) -> "ScanParams":
        """MotionScanner::initialize() lines 184-199; None -> the env/default getter."""
        lib = load_library()
        c = ScanParamsC()
        check(lib.mtgpu_params_from_config(
            C.byref(c), int(width), int(height),
            config.mv_threshold_sq() if mv_threshold_sq is None else float(mv_threshold_sq),
            config.block_size() if block_size is None else int(block_size),
            config.block_shift() if block_shift is None else int(block_shift),
            config.vectors_needed() if vectors_needed is None else int(vectors_needed),
            config.clusters_needed() if clusters_needed is None else int(clusters_needed),
            config.vertical_mask() if vertical_mask is None else float(vertical_mask)))
        return cls.from_c(c)

    @classmethod
    def from_c(cls, c: ScanParamsC) -> "ScanParams":
        return cls(c.mv_threshold_sq, c.block_shift, c.clusters_needed, c.vertical_margin,
                   c.vectors_needed, c.grid_w, c.grid_h)

    def to_c(self) -> ScanParamsC:
        c = ScanParamsC()
        c.mv_threshold_sq = self.mv_threshold_sq
        c.block_shift = self.block_shift
        c.clusters_needed = self.clusters_needed
        c.vertical_margin = self.vertical_margin
        c.vectors_needed = self.vectors_needed & 0xFF
        c.grid_w = self.grid_w
        c.grid_h = self.grid_h
        return c


@dataclass
class MergeParams:
    """mt_merge_params: MAX_GAP_SEC, PADDING_SEC, duration, MIN_SAVINGS_PCT."""
    duration: float
    max_gap_sec: float = None
    padding_sec: float = None
    min_savings_pct: float = None

    def __post_init__(self):
        if self.max_gap_sec is None:
            self.max_gap_sec = config.max_gap_sec()
        if self.padding_sec is None:
            self.padding_sec = config.padding_sec()
        if self.min_savings_pct is None:
            self.min_savings_pct = config.min_savings_pct()

    def to_c(self) -> MergeParamsC:
        return MergeParamsC(self.max_gap_sec, self.padding_sec, self.duration, self.min_savings_pct)

    def to_record(self) -> np.ndarray:
        return np.array([(self.max_gap_sec, self.padding_sec, self.duration, self.min_savings_pct)],
                        dtype=MERGE_PARAMS_DTYPE)


@dataclass
class FrameBatch:
    """CSR batch of extracted MV side data: frame f owns mv[frame_off[f]:frame_off[f+1]]."""
    mv: np.ndarray                      # MV_DTYPE, packed
    frame_off: np.ndarray               # uint64 [F+1]
    pts: Optional[np.ndarray] = None    # float64 [F], seconds
    has_sd: Optional[np.ndarray] = None  # uint8 [F]; None: side data iff >= 1 record

    @property
    def n_frames(self) -> int:
        return len(self.frame_off) - 1

    @classmethod
    def from_frames(cls, frames: Sequence[Optional[np.ndarray]], pts=None) -> "FrameBatch":
        """frames[i] is an MV_DTYPE array, or None when the frame had no MV side data."""
        counts = [0 if f is None else len(f) for f in frames]
        off = np.zeros(len(frames) + 1, dtype=np.uint64)
        np.cumsum(counts, out=off[1:])
        mv = np.zeros(int(off[-1]), dtype=MV_DTYPE)
        for i, f in enumerate(frames):
            if f is not None and len(f):
                mv[int(off[i]):int(off[i + 1])] = f
        has_sd = np.array([0 if f is None else 1 for f in frames], dtype=np.uint8)
        return cls(mv, off, None if pts is None else np.asarray(pts, dtype=np.float64), has_sd)


def frame_skip(video_fps: float, target_fps: float) -> int:
    """src/motion_scanner.cpp:311-313."""
    return int(video_fps / target_fps) if (target_fps > 0 and target_fps < video_fps) else 1


def filter_frames(frame_pts: Sequence[int], time_base: float, start: float, end: float,
                  skip: int) -> Tuple[List[int], List[float]]:
    """Which decoded frames of one scan_range(start, end) call reach check_frame
    (src/motion_scanner.cpp:314, 357-371).  Returns (indices, pts_seconds)."""
    idx, pts_out = [], []
    count = 0
    for i, p in enumerate(frame_pts):
        count += 1
        if count % skip != 0:
            continue
        pts = float(p) * time_base
        if pts < start:
            continue
        if pts >= end:
            break
        idx.append(i)
        pts_out.append(pts)
    return idx, pts_out


def make_chunks(duration: float, chunk_sec: Optional[float] = None) -> List[Tuple[float, float, int]]:
    """src/pipeline.cpp:163-167: ScanTask{start, end, id}."""
    chunk_sec = config.chunk_duration_sec() if chunk_sec is None else chunk_sec
    out, t, cid = [], 0.0, 0
    while t < duration:
        out.append((t, min(t + chunk_sec, duration), cid))
        cid += 1
        t += chunk_sec
    return out


def concat_list(segments, abs_input_path: str) -> str:
    """The text the reference's cut executor feeds to `ffmpeg -f concat` for a job's segments
    (src/ffmpeg_executor.cpp:38-50): per segment with end > start three lines — file '<path>', inpoint and
    outpoint with two decimals (fmt's {:.2f} == "%.2f").  Convenience for diffing against the reference's cut
    list; the hand-off itself is the segment doubles."""
    out = []
    for s in segments:
        a, b = float(s[0]), float(s[1])
        if b <= a:
            continue
        out.append("file '%s'\ninpoint %.2f\noutpoint %.2f\n" % (abs_input_path, a, b))
    return "".join(out)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def plan_preview(params: "ScanParams", lds_bytes: int = 163840, cu_count: int = 256) -> dict:
    """The launch plan the library would pick for `params` on a device with that much LDS per
    workgroup and that many CUs (defaults: MI355X).  Host arithmetic only: works without a GPU."""
    p = PlanC()
    c = params.to_c()
    check(load_library().mtgpu_plan_preview(C.byref(c), int(lds_bytes), int(cu_count), C.byref(p)))
    return {n: getattr(p, n) for n, _ in PlanC._fields_ if not n.startswith("_")}


def pack_records(mv: np.ndarray) -> np.ndarray:
    """40-byte AVMotionVector records -> the 8-byte compact records the host dispatcher stages
    (bytes 6..13: src_x, src_y, dst_x, dst_y) through the library's own packer (mtgpu_pack_records)."""
    mv = np.ascontiguousarray(mv, dtype=MV_DTYPE)
    out = np.zeros(len(mv), dtype=COMPACT_DTYPE)
    if len(mv):
        check(load_library().mtgpu_pack_records(_ptr(mv), len(mv), _ptr(out)))
    return out


class MotionScanner:
    """GPU scanner context (one per device; callable from many threads)."""

    def __init__(self, params: ScanParams, device: int = 0):
        self.params = params
        self.device = device
        self._lib = load_library()
        self._ctx = C.c_void_p()
        c = params.to_c()
        check(self._lib.mtgpu_create(C.byref(c), int(device), C.byref(self._ctx)))

    # -- reference-shaped constructor: MotionScanner(...).initialize()
    @classmethod
    def initialize(cls, width: int, height: int, device: int = 0, **cfg) -> "MotionScanner":
        return cls(ScanParams.from_config(width, height, **cfg), device)

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.mtgpu_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def plan(self) -> dict:
        p = PlanC()
        check(self._lib.mtgpu_get_plan(self._ctx, C.byref(p)))
        return {n: getattr(p, n) for n, _ in PlanC._fields_ if not n.startswith("_")}

    def set_slices(self, slices: int):
        """Workgroups per frame: 0 = automatic (default), or 1 / 2 / 4 / 8.  Never changes results."""
        check(self._lib.mtgpu_set_slices(self._ctx, int(slices)))

    def profile(self, on: bool = True):
        """Launch timing on / off (mtgpu_profile_enable): three events per scan launch, read with profile_read()."""
        check(self._lib.mtgpu_profile_enable(self._ctx, 1 if on else 0))

    def profile_read(self) -> dict:
        """Waits for the profiled launches so far; {"plan_ms", "scan_ms"}: MEAN per launch over "launches" launches."""
        a, b, n = C.c_double(), C.c_double(), C.c_uint32()
        check(self._lib.mtgpu_profile_read(self._ctx, C.byref(a), C.byref(b), C.byref(n)))
        k = max(n.value, 1)
        return {"plan_ms": a.value / k, "scan_ms": b.value / k, "launches": n.value}

    def stats(self) -> dict:
        """What the context holds on the device (mtgpu_get_stats): staging of the host-pointer entry points,
        the scratch pool's reserved bytes now and at its high-water mark."""
        from ._abi import CtxStatsC
        st = CtxStatsC()
        check(self._lib.mtgpu_get_stats(self._ctx, C.byref(st)))
        return {n: getattr(st, n) for n, _ in CtxStatsC._fields_}

    def trim(self):
        """Hand the scratch pool's unused blocks back to the device (mtgpu_trim)."""
        check(self._lib.mtgpu_trim(self._ctx))

    # ---------------------------------------------------------------- scan
    def check_frames(self, batch: FrameBatch) -> np.ndarray:
        """check_frame() for every frame of a host batch -> uint8 flags [F]."""
        mv = np.ascontiguousarray(batch.mv, dtype=MV_DTYPE)
        off = np.ascontiguousarray(batch.frame_off, dtype=np.uint64)
        sd = None if batch.has_sd is None else np.ascontiguousarray(batch.has_sd, dtype=np.uint8)
        n = len(off) - 1
        flags = np.zeros(max(n, 0), dtype=np.uint8)
        if n <= 0:
            return flags
        check(self._lib.mtgpu_scan_frames(self._ctx, _ptr(mv) if len(mv) else None, _ptr(off),
                                          _ptr(sd), n, _ptr(flags)))
        return flags

    def check_frames_device(self, mv, frame_off, has_sd=None, flags=None, stream=None):
        """Device-resident batch (torch CUDA tensors).  mv: uint8 [n_records*40] (or any
        dtype viewing the packed records), frame_off: int64 [F+1].  Asynchronous on
        `stream` (default: torch's current stream).  Returns the uint8 flags tensor."""
        import torch
        n_frames = frame_off.numel() - 1
        if flags is None:
            flags = torch.empty(max(n_frames, 0), dtype=torch.uint8, device=frame_off.device)
        if n_frames <= 0:
            return flags
        assert mv.is_contiguous() and frame_off.is_contiguous() and flags.is_contiguous()
        assert frame_off.dtype == torch.int64 and flags.dtype == torch.uint8
        n_records = (mv.numel() * mv.element_size()) // 40
        st = torch.cuda.current_stream(frame_off.device).cuda_stream if stream is None else stream
        check(self._lib.mtgpu_scan_frames_device(
            self._ctx, mv.data_ptr() if n_records else None, n_records, frame_off.data_ptr(),
            None if has_sd is None else has_sd.data_ptr(), n_frames, flags.data_ptr(), st))
        return flags

    def check_frames_device_compact(self, rec8, frame_off, has_sd=None, flags=None, stream=None):
        """Device-resident batch of COMPACT records (8 bytes each, see pack_records): rec8 is a
        torch CUDA tensor viewing n_records * 8 bytes.  Otherwise like check_frames_device."""
        import torch
        n_frames = frame_off.numel() - 1
        if flags is None:
            flags = torch.empty(max(n_frames, 0), dtype=torch.uint8, device=frame_off.device)
        if n_frames <= 0:
            return flags
        assert rec8.is_contiguous() and frame_off.is_contiguous() and flags.is_contiguous()
        assert frame_off.dtype == torch.int64 and flags.dtype == torch.uint8
        n_records = (rec8.numel() * rec8.element_size()) // 8
        st = torch.cuda.current_stream(frame_off.device).cuda_stream if stream is None else stream
        check(self._lib.mtgpu_scan_frames_device_compact(
            self._ctx, rec8.data_ptr() if n_records else None, n_records, frame_off.data_ptr(),
            None if has_sd is None else has_sd.data_ptr(), n_frames, flags.data_ptr(), st))
        return flags

    def scan_range(self, frame_pts: Sequence[int], frames: Sequence[Optional[np.ndarray]],
                   time_base: float, start: float, end: float, video_fps: float,
                   target_fps: Optional[float] = None) -> List[float]:
        """MotionScanner::scan_range (src/motion_scanner.cpp:297-391) over frames that a
        host decoder already produced for this chunk: `frame_pts[i]` / `frames[i]` are the
        AVFrame::pts and the MV side data (None = no side data) of the i-th frame returned
        after the seek.  Returns the motion timestamps of [start, end)."""
        skip = frame_skip(video_fps, config.target_fps() if target_fps is None else target_fps)
        idx, pts = filter_frames(frame_pts, time_base, start, end, skip)
        if not idx:
            return []
        batch = FrameBatch.from_frames([frames[i] for i in idx], pts)
        flags = self.check_frames(batch)
        return [p for p, f in zip(pts, flags) if f]

    # --------------------------------------------------------------- merge
    def merge_segments(self, timestamps: Iterable[float], mp: MergeParams, job_semantics: bool = False,
                       cap: Optional[int] = None) -> Tuple[np.ndarray, dict]:
        """sort + unique + gap merge + clamp + savings + cut decision on the device
        (src/pipeline.cpp:302-358, 387-388).  Returns (segments, result)."""
        ts = np.ascontiguousarray(np.asarray(list(timestamps) if not isinstance(timestamps, np.ndarray)
                                             else timestamps, dtype=np.float64))
        cap = (len(ts) + 1) if cap is None else cap
        seg = np.zeros(cap, dtype=SEGMENT_DTYPE)
        res = MergeResultC()
        c_mp = mp.to_c()
        check(self._lib.mtgpu_merge_segments(self._ctx, _ptr(ts) if len(ts) else None, len(ts),
                                             C.byref(c_mp), 1 if job_semantics else 0,
                                             _ptr(seg) if cap else None, cap, C.byref(res)))
        out = {n: getattr(res, n) for n, _ in MergeResultC._fields_}
        return seg[:res.n_segments].copy(), out

    def merge_timestamps_device(self, ts, mp: MergeParams, job_semantics: bool = False, seg_cap: int = 64,
                                stream=None):
        """The same merge for ONE stream's pooled timestamps already on the device (1-D float64
        CUDA tensor, any order, duplicates allowed); large lists are sorted / merged by many
        workgroups.  Returns (segments float64 [seg_cap, 2], result uint8 [40]) device tensors;
        asynchronous on `stream` (default: torch's current stream)."""
        import torch
        assert ts.dtype == torch.float64 and ts.is_contiguous()
        dev = ts.device
        seg = torch.zeros((seg_cap, 2), dtype=torch.float64, device=dev)
        res = torch.zeros(MERGE_RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        c_mp = mp.to_c()
        check(self._lib.mtgpu_merge_timestamps_device(
            self._ctx, ts.data_ptr() if ts.numel() else None, ts.numel(), C.byref(c_mp),
            1 if job_semantics else 0, seg.data_ptr(), seg_cap, res.data_ptr(), st))
        return seg, res

    def merge_streams_device(self, flags, pts, stream_off, merge_params, job_semantics=False,
                             seg_cap=64, stream=None, out=None):
        """Per-stream timestamp pooling + merge without leaving the device.
        flags uint8 [F], pts float64 [F], stream_off int64 [S+1], merge_params: uint8 view of
        S mt_merge_params records (torch tensors on the scanner's device).
        Returns (segments float64 [S, seg_cap, 2], results uint8 [S, 40]).
        `out` = (segments, results, workspace float64 [2F]) reuses caller buffers (no allocation,
        no zero fill: entries past n_segments are then unspecified)."""
        import torch
        dev = pts.device
        n_streams = stream_off.numel() - 1
        n_frames = pts.numel()
        if out is not None:
            seg, res, ws = out
            assert seg.is_contiguous() and res.is_contiguous() and ws.numel() >= 2 * n_frames
            assert tuple(seg.shape) == (n_streams, seg_cap, 2) and res.shape[0] == n_streams
        else:
            ws = torch.empty(2 * max(n_frames, 1), dtype=torch.float64, device=dev)
            seg = torch.zeros((max(n_streams, 0), seg_cap, 2), dtype=torch.float64, device=dev)
            res = torch.zeros((max(n_streams, 0), MERGE_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        if n_streams <= 0:
            return seg, res
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        check(self._lib.mtgpu_merge_streams_device(
            self._ctx, None if flags is None else flags.data_ptr(), pts.data_ptr(),
            stream_off.data_ptr(), n_streams, merge_params.data_ptr(), 1 if job_semantics else 0,
            ws.data_ptr(), seg.data_ptr(), seg_cap, res.data_ptr(), st))
        if out is None:
            # the workspace must outlive the kernel that was just queued: it is held until an event recorded
            # behind the launch on ITS stream has passed (torch's caching allocator knows nothing of a raw
            # hipStream_t handed in by the caller, and several merges may be queued back to back)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev) if stream is None else torch.cuda.ExternalStream(stream, device=dev))
            self._held = [(e, w) for e, w in getattr(self, "_held", []) if not e.query()]
            self._held.append((ev, ws))
        return seg, res


class ScanPipe:
    """Pinned, multi-buffered host->device scan pipeline (include/mtgpu.h "Host dispatcher"):
    what a decoder thread uses instead of calling check_frame per frame
    (src/motion_scanner.cpp:375-383).  feed() copies a frame's MV side data into pinned
    staging and submits full batches asynchronously; drain() returns every finished
    (pts, flag, tag) in submission order."""

    def __init__(self, scanner: MotionScanner, max_records: int, max_frames: int, n_buffers: int = 3,
                 layout: int = LAYOUT_COMPACT8 | LAYOUT_ZERO_COPY):
        """layout: LAYOUT_COMPACT8 (8 of every 40 record bytes are staged) or LAYOUT_AOS40 (records
        staged unchanged), optionally | LAYOUT_ZERO_COPY (the scan reads the pinned staging over
        PCIe itself: no copy commands).  Default: compact + zero-copy.  Results are identical."""
        self._lib = scanner._lib
        self._scanner = scanner            # keeps the context alive
        self._pipe = C.c_void_p()
        check(self._lib.mtgpu_pipe_create_layout(scanner._ctx, int(max_records), int(max_frames),
                                                 int(n_buffers), int(layout), C.byref(self._pipe)))
        self._cur = None
        self._inflight = 0
        self._done: List[Tuple[float, int, int]] = []

    def close(self):
        if getattr(self, "_pipe", None) and self._pipe.value:
            self._lib.mtgpu_pipe_destroy(self._pipe)
            self._pipe = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _collect_one(self):
        b, fl, pts, tags, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
        rc = self._lib.mtgpu_pipe_collect(self._pipe, C.byref(b), C.byref(fl), C.byref(pts), C.byref(tags),
                                          C.byref(n))
        if rc != _abi.MT_OK:
            msg = self._lib.mtgpu_last_error().decode("utf-8", "replace")
            if b.value:                         # a failed collect still hands the batch out: give it back
                self._lib.mtgpu_pipe_release(self._pipe, b)
                self._inflight -= 1
            raise _abi.MtgpuError(rc, msg)
        k = n.value
        if k:
            f = np.ctypeslib.as_array(C.cast(fl, C.POINTER(C.c_uint8)), (k,)).copy()
            p = np.ctypeslib.as_array(C.cast(pts, C.POINTER(C.c_double)), (k,)).copy()
            t = np.ctypeslib.as_array(C.cast(tags, C.POINTER(C.c_uint64)), (k,)).copy()
            self._done += list(zip(p.tolist(), f.tolist(), t.tolist()))
        check(self._lib.mtgpu_pipe_release(self._pipe, b))
        self._inflight -= 1

    def _acquire(self):
        while True:
            b = C.c_void_p()
            rc = self._lib.mtgpu_pipe_acquire(self._pipe, C.byref(b))
            if rc == _abi.MT_ERR_BUSY:          # back-pressure: finish the oldest batch first
                self._collect_one()
                continue
            check(rc)
            return b

    def _submit(self):
        if self._cur is not None:
            check(self._lib.mtgpu_pipe_submit(self._pipe, self._cur))
            self._cur = None
            self._inflight += 1

    def feed(self, mv: Optional[np.ndarray], pts: float, tag: int = 0):
        """One decoded frame: `mv` is its MV side data (MV_DTYPE array or raw bytes), None when
        the frame has no side data."""
        if mv is None:
            ptr, nbytes, sd = None, 0, 0
        else:
            a = np.ascontiguousarray(mv)
            ptr, nbytes, sd = (a.ctypes.data_as(C.c_void_p) if a.nbytes else None), a.nbytes, 1
        while True:
            if self._cur is None:
                self._cur = self._acquire()
            rc = self._lib.mtgpu_batch_add_frame(self._cur, ptr, nbytes, sd, float(pts), int(tag))
            if rc == _abi.MT_ERR_CAPACITY:      # batch full: ship it, start the next one
                self._submit()
                continue
            check(rc)
            return

    def drain(self) -> List[Tuple[float, int, int]]:
        """Submit the partial batch, wait for everything in flight, return and clear the
        accumulated (pts, flag, tag) list."""
        if self._cur is not None and self._lib.mtgpu_batch_frames(self._cur) > 0:
            self._submit()
        elif self._cur is not None:
            check(self._lib.mtgpu_pipe_release(self._pipe, self._cur))
            self._cur = None
        while self._inflight:
            self._collect_one()
        out, self._done = self._done, []
        return out


def results_from_bytes(res_bytes: np.ndarray) -> np.ndarray:
    """uint8 [S,40] (host) -> structured mt_merge_result records."""
    return np.ascontiguousarray(res_bytes).view(MERGE_RESULT_DTYPE).reshape(-1)
