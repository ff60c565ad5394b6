"""MI355X-native motion-vector scanner: the hot path of
Vaibhav-20022002/Motion-Estimated-Video-Trimmer (MotionScanner::check_frame + the
gap-bounded segment merge) as hand-written gfx950 HIP kernels behind the C ABI of
include/mtgpu.h.  This package is the thin host-side mirror of the reference's
scanner interface; the compute lives in libmtgpu.so (csrc/)."""
from . import config, mvfile, mvjson
from ._abi import (COMPACT_DTYPE, LAYOUT_AOS40, LAYOUT_COMPACT8, LAYOUT_ZERO_COPY, LIB_PATH, MERGE_PARAMS_DTYPE, MERGE_RESULT_DTYPE, MV_DTYPE, SEGMENT_DTYPE,
                   MtgpuError, load_library)
from .scanner import (FrameBatch, MergeParams, MotionScanner, ScanParams, ScanPipe, concat_list, filter_frames,
                      frame_skip, make_chunks, pack_records, plan_preview, results_from_bytes)

__all__ = ["config", "mvfile", "mvjson", "LIB_PATH", "COMPACT_DTYPE", "LAYOUT_AOS40", "LAYOUT_COMPACT8", "LAYOUT_ZERO_COPY",
           "pack_records", "plan_preview", "MV_DTYPE", "SEGMENT_DTYPE", "MERGE_PARAMS_DTYPE",
           "MERGE_RESULT_DTYPE", "MtgpuError", "load_library", "FrameBatch", "MergeParams",
           "MotionScanner", "ScanParams", "ScanPipe", "concat_list", "filter_frames", "frame_skip", "make_chunks",
           "results_from_bytes"]
