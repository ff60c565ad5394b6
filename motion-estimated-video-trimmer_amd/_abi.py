"""ctypes view of the C ABI declared in include/mtgpu.h / include/mt_types.h.

The product path is the HIP library ``libmtgpu.so`` built in-tree by
``csrc/Makefile``.  There is no Python or CPU fallback: if the library is missing
``load_library()`` raises, and every compute entry point of the library itself
fails with MT_ERR_DEVICE when no gfx950 device is usable.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libmtgpu.so")

LAYOUT_COMPACT8, LAYOUT_AOS40, LAYOUT_ZERO_COPY = 0, 1, 2
COMPACT_DTYPE = np.dtype([("src_x", "<i2"), ("src_y", "<i2"), ("dst_x", "<i2"), ("dst_y", "<i2")])
MT_OK, MT_ERR_INVALID, MT_ERR_CAPACITY, MT_ERR_DEVICE, MT_ERR_NOMEM, MT_ERR_BUSY, MT_ERR_UNSUPPORTED = 0, 1, 2, 3, 4, 5, 6
# copy-out loops of mtgpu_pack_records_with (include/mtgpu.h)
PACK_SCALAR, PACK_AVX2, PACK_AVX512, PACK_IMPL_MASK, PACK_NT = 1, 2, 3, 15, 16

# AVMotionVector-compatible record (include/mt_types.h: mt_mv; 40 bytes).
MV_DTYPE = np.dtype(
    {
        "names": ["source", "w", "h", "src_x", "src_y", "dst_x", "dst_y", "flags",
                  "motion_x", "motion_y", "motion_scale"],
        "formats": ["<i4", "u1", "u1", "<i2", "<i2", "<i2", "<i2", "<u8", "<i4", "<i4", "<u2"],
        "offsets": [0, 4, 5, 6, 8, 10, 12, 16, 24, 28, 32],
        "itemsize": 40,
    }
)
SEGMENT_DTYPE = np.dtype([("start", "<f8"), ("end", "<f8")])
MERGE_PARAMS_DTYPE = np.dtype([("max_gap_sec", "<f8"), ("padding_sec", "<f8"),
                               ("duration", "<f8"), ("min_savings_pct", "<f8")])
MERGE_RESULT_DTYPE = np.dtype([("n_timestamps", "<u8"), ("n_segments", "<u8"),
                               ("time_removed", "<f8"), ("saved_pct", "<f8"),
                               ("do_cut", "<i4"), ("status", "<i4")])


class ScanParamsC(C.Structure):
    _fields_ = [("mv_threshold_sq", C.c_double), ("block_shift", C.c_int32),
                ("clusters_needed", C.c_int32), ("vertical_margin", C.c_int32),
                ("vectors_needed", C.c_uint8), ("_pad", C.c_uint8 * 3),
                ("grid_w", C.c_int32), ("grid_h", C.c_int32)]


class MergeParamsC(C.Structure):
    _fields_ = [("max_gap_sec", C.c_double), ("padding_sec", C.c_double),
                ("duration", C.c_double), ("min_savings_pct", C.c_double)]


class MergeResultC(C.Structure):
    _fields_ = [("n_timestamps", C.c_uint64), ("n_segments", C.c_uint64),
                ("time_removed", C.c_double), ("saved_pct", C.c_double),
                ("do_cut", C.c_int32), ("status", C.c_int32)]


class PlanC(C.Structure):
    _fields_ = [("block_threads", C.c_int32), ("bands", C.c_int32), ("band_rows", C.c_int32),
                ("lds_bytes", C.c_int32), ("counter_bits", C.c_int32), ("device", C.c_int32),
                ("cu_count", C.c_int32), ("chunk_rows", C.c_int32),
                ("counter_mode", C.c_int32), ("_pad", C.c_int32)]


class CtxStatsC(C.Structure):
    _fields_ = [("staging_device_bytes", C.c_uint64), ("pool_reserved_bytes", C.c_uint64),
                ("pool_reserved_high", C.c_uint64), ("hip_streams", C.c_uint32), ("private_pool", C.c_uint32)]


class PipeStatsC(C.Structure):
    _fields_ = [("pinned_bytes", C.c_uint64), ("device_bytes", C.c_uint64), ("submits", C.c_uint64),
                ("n_buffers", C.c_uint32), ("layout", C.c_int32), ("pin_us", C.c_uint64),
                ("pinned_batches", C.c_uint32), ("hip_streams", C.c_uint32), ("list_bytes", C.c_uint64)]


assert C.sizeof(ScanParamsC) == 32 and C.sizeof(MergeResultC) == 40

# name -> (restype, argtypes): every symbol include/mtgpu.h declares.
ABI = {
    "mtgpu_version": (C.c_char_p, []),
    "mtgpu_last_error": (C.c_char_p, []),
    "mtgpu_device_count": (C.c_int, []),
    "mtgpu_device_pci_address": (C.c_int, [C.c_int, C.c_char_p, C.c_uint64]),
    "mtgpu_params_from_config": (C.c_int, [C.POINTER(ScanParamsC), C.c_int, C.c_int, C.c_double,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "mtgpu_create": (C.c_int, [C.POINTER(ScanParamsC), C.c_int, C.POINTER(C.c_void_p)]),
    "mtgpu_destroy": (None, [C.c_void_p]),
    "mtgpu_get_params": (C.c_int, [C.c_void_p, C.POINTER(ScanParamsC)]),
    "mtgpu_get_stats": (C.c_int, [C.c_void_p, C.POINTER(CtxStatsC)]),
    "mtgpu_trim": (C.c_int, [C.c_void_p]),
    "mtgpu_get_plan": (C.c_int, [C.c_void_p, C.POINTER(PlanC)]),
    "mtgpu_plan_preview": (C.c_int, [C.POINTER(ScanParamsC), C.c_int, C.c_int, C.POINTER(PlanC)]),
    "mtgpu_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "mtgpu_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "mtgpu_set_slices": (C.c_int, [C.c_void_p, C.c_int]),
    "mtgpu_scan_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                           C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mtgpu_scan_frames_device_compact": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                                   C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mtgpu_pack_records": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "mtgpu_pack_records_with": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mtgpu_pack_selected": (C.c_int, []),
    "mtgpu_scan_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint32, C.c_void_p]),
    "mtgpu_merge_segments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(MergeParamsC),
                                       C.c_int, C.c_void_p, C.c_uint64, C.POINTER(MergeResultC)]),
    "mtgpu_merge_timestamps_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(MergeParamsC),
                                                C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mtgpu_merge_streams_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_uint32, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "mtgpu_pipe_create": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "mtgpu_pipe_create_layout": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p)]),
    "mtgpu_pipe_destroy": (None, [C.c_void_p]),
    "mtgpu_pipe_acquire": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mtgpu_batch_add_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_double, C.c_uint64]),
    "mtgpu_batch_frames": (C.c_uint32, [C.c_void_p]),
    "mtgpu_pipe_submit": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtgpu_pipe_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]),
    "mtgpu_pipe_release": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtgpu_pipe_get_stats": (C.c_int, [C.c_void_p, C.POINTER(PipeStatsC)]),
    "mtgpu_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mtgpu_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mtgpu_comm_destroy": (None, [C.c_void_p]),
    "mtgpu_gather_segments": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
}

_lib = None


class MtgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mtgpu error {code}: {msg}")
        self.code = code


def load_library(path=None):
    """Load libmtgpu.so (built by csrc/Makefile).  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # MTGPU_LIBRARY: another build of the same ABI (developers: the experiments build, csrc/Makefile `experiments`)
    p = path or os.environ.get("MTGPU_LIBRARY") or LIB_PATH
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7 /
    # libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm).  If libmtgpu.so pulled in the
    # system copies first, a later `import torch` would find "No HIP GPUs".  Importing
    # torch first makes both share torch's runtime; without torch the system ROCm is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise FileNotFoundError(
            f"{p} not found: build it with `make -C {os.path.join(PKG_DIR, 'csrc')}` "
            "(or __graft_entry__.build()).  There is no fallback path.")
    lib = C.CDLL(p)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(rc):
    if rc != MT_OK:
        raise MtgpuError(rc, load_library().mtgpu_last_error().decode("utf-8", "replace"))
