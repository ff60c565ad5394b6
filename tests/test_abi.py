"""The C-ABI library builds, loads and exports every symbol include/mtgpu.h declares.
No compute calls (there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import mvtrim_amd as m
from mvtrim_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mtgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtgpu_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = m.load_library()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mtgpu.h but not exported by libmtgpu.so"
        assert n in _abi.ABI, f"{n} has no ctypes prototype in _abi.py"
    assert sorted(_abi.ABI) == names
    assert b"mtgpu" in lib.mtgpu_version()


def test_product_library_does_not_link_the_oracle():
    """The product must never route through the oracle or any CPU fallback."""
    out = subprocess.check_output(["ldd", _abi.LIB_PATH]).decode()
    assert "mt_oracle" not in out and "libamdhip64" in out
    syms = subprocess.check_output(["nm", "-D", "--defined-only", _abi.LIB_PATH]).decode()
    assert "mto_" not in syms
    # kernels for gfx950 are embedded
    blob = open(_abi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"scan_frames_kernel" in blob and b"merge_streams_kernel" in blob
    for f in os.listdir(os.path.join(ROOT, "motion-estimated-video-trimmer_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "motion-estimated-video-trimmer_amd", f)).read()
            assert "oracle" not in src.lower(), f"{f} mentions the oracle"


def test_product_library_carries_no_calibration_code():
    """bench.py's read-only calibration kernels live in a library of their own (csrc/calib -> libmtgpu_calib.so): the
    product exports nothing named debug / calib, embeds no read_ceiling kernel, and neither it nor the Python package
    refers to the calibration library; that library in turn exports exactly what csrc/calib/mtgpu_calib.h declares."""
    import re
    syms = subprocess.check_output(["nm", "-D", "--defined-only", _abi.LIB_PATH]).decode()
    assert "debug" not in syms and "calib" not in syms
    blob = open(_abi.LIB_PATH, "rb").read()
    assert b"read_ceiling" not in blob and b"libmtgpu_calib" not in blob
    assert b"plan_scatter_kernel" in blob and b"plan_count_kernel" in blob          # the work list's kernels are product
    pkg = os.path.join(ROOT, "motion-estimated-video-trimmer_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "calib" not in open(os.path.join(pkg, f)).read(), f
    calib = os.path.join(pkg, "libmtgpu_calib.so")
    assert os.path.exists(calib), "make -C csrc all builds it (bench.py loads it)"
    hdr = open(os.path.join(pkg, "csrc", "calib", "mtgpu_calib.h")).read()
    declared = sorted(set(re.findall(r"\b(mtcalib_\w+)\s*\(", hdr)))
    exported = sorted(ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", calib]).decode().splitlines()
                      if " T " in ln and "mtcalib_" in ln)
    assert declared == exported == ["mtcalib_last_error", "mtcalib_read_ceiling"]


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.ScanParamsC) == 32 and _abi.ScanParamsC.grid_w.offset == 24
    assert _abi.ScanParamsC.vectors_needed.offset == 20 and _abi.ScanParamsC.mv_threshold_sq.offset == 0
    assert C.sizeof(_abi.MergeParamsC) == 32 and C.sizeof(_abi.MergeResultC) == 40
    assert _abi.MERGE_RESULT_DTYPE.itemsize == 40 and _abi.SEGMENT_DTYPE.itemsize == 16
    assert _abi.MV_DTYPE.itemsize == 40 and _abi.MV_DTYPE.fields["dst_y"][1] == 12
    # the same header compiles as C and as C++ with its static_asserts
    for comp, flag in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        subprocess.check_call([comp, flag, "-fsyntax-only", "-x", "c" if comp == "gcc" else "c++",
                               os.path.join(ROOT, "include", "mtgpu.h")])


def test_no_device_means_loud_failure_not_fallback():
    lib = m.load_library()
    if lib.mtgpu_device_count() > 0:
        pytest.skip("a GPU is present; the no-device failure path is for the CPU tier")
    with pytest.raises(m.MtgpuError) as ei:
        m.MotionScanner(m.ScanParams.from_config(1920, 1080))
    assert ei.value.code == _abi.MT_ERR_DEVICE and "no CPU fallback" in str(ei.value)


def test_invalid_arguments_are_reported():
    lib = m.load_library()
    c = _abi.ScanParamsC()
    assert lib.mtgpu_params_from_config(C.byref(c), 1920, 1080, 16.0, 16, 40, 2, 2, 0.05) == _abi.MT_ERR_INVALID
    assert b"block_shift" in lib.mtgpu_last_error()
    assert lib.mtgpu_params_from_config(None, 1920, 1080, 16.0, 16, 4, 2, 2, 0.05) == _abi.MT_ERR_INVALID
    assert lib.mtgpu_params_from_config(C.byref(c), 10 ** 7, 1080, 16.0, 16, 4, 2, 2, 0.05) == _abi.MT_ERR_INVALID
    ctx = C.c_void_p()
    bad = m.ScanParams.from_config(1920, 1080).to_c()
    bad.grid_w = 0
    assert lib.mtgpu_create(C.byref(bad), 0, C.byref(ctx)) == _abi.MT_ERR_INVALID


def test_null_and_foreign_handles_are_rejected_without_a_device():
    """Argument checks of every handle-taking entry point run before any HIP call: NULL contexts,
    pipes, batches and communicators are MT_ERR_INVALID (never a crash), on a box with no GPU too."""
    lib = m.load_library()
    inv = _abi.MT_ERR_INVALID
    null = C.c_void_p()
    out = C.c_void_p()
    mp = _abi.MergeParamsC(5.0, 0.5, 10.0, 5.0)
    res = _abi.MergeResultC()
    assert lib.mtgpu_get_plan(None, None) == inv and lib.mtgpu_get_params(None, None) == inv
    assert lib.mtgpu_set_slices(None, 2) == inv
    assert lib.mtgpu_scan_frames(None, None, None, None, 1, None) == inv
    assert lib.mtgpu_scan_frames_device(None, None, 0, None, None, 1, None, None) == inv
    assert lib.mtgpu_scan_frames_device_compact(None, None, 0, None, None, 1, None, None) == inv
    assert lib.mtgpu_merge_segments(None, None, 0, C.byref(mp), 0, None, 0, C.byref(res)) == inv
    assert lib.mtgpu_merge_timestamps_device(None, None, 0, C.byref(mp), 0, None, 0, None, None) == inv
    assert lib.mtgpu_merge_streams_device(None, None, None, None, 1, None, 0, None, None, 0, None, None) == inv
    assert lib.mtgpu_pipe_create(None, 100, 4, 2, C.byref(out)) == inv and not out.value
    assert lib.mtgpu_pipe_create_layout(None, 100, 4, 2, 0, C.byref(out)) == inv
    assert lib.mtgpu_pipe_acquire(None, C.byref(out)) == inv
    assert lib.mtgpu_batch_add_frame(None, None, 0, 0, 0.0, 0) == inv and lib.mtgpu_batch_frames(None) == 0
    assert lib.mtgpu_pipe_submit(None, None) == inv and lib.mtgpu_pipe_release(None, None) == inv
    assert lib.mtgpu_pipe_collect(None, C.byref(out), None, None, None, None) == inv
    assert lib.mtgpu_gather_segments(None, None, 16, None, None) == inv
    assert lib.mtgpu_comm_create(0, 0, None, 0, C.byref(out)) == inv
    assert lib.mtgpu_pack_records(None, 3, None) == inv and lib.mtgpu_pack_records(None, 0, None) == _abi.MT_OK
    lib.mtgpu_pipe_destroy(None)
    lib.mtgpu_destroy(None)
    lib.mtgpu_comm_destroy(None)
    del null


def test_pack_records_known_answer():
    """mtgpu_pack_records keeps bytes 6..13 of every 40-byte record, in order (host data movement)."""
    import numpy as np
    mv = np.zeros(3, dtype=m.MV_DTYPE)
    mv["src_x"], mv["src_y"], mv["dst_x"], mv["dst_y"] = [1, -2, 32767], [3, -4, -32768], [5, -6, 7], [8, -9, 10]
    mv["w"], mv["h"], mv["flags"], mv["motion_x"] = 255, 255, 2 ** 63, -1          # neighbours of the kept bytes
    raw = mv.view(np.uint8).reshape(3, 40)
    raw[:, 14:16] = 0xAB                                                            # padding after dst_y
    got = m.pack_records(mv)
    assert got.dtype == m.COMPACT_DTYPE and got.itemsize == 8
    assert got["src_x"].tolist() == [1, -2, 32767] and got["dst_y"].tolist() == [8, -9, 10]
    assert got.view(np.uint8).reshape(3, 8).tobytes() == raw[:, 6:14].tobytes()


def test_every_environment_knob_is_in_the_headers_table():
    """Every MTGPU_* variable the library and the C++ host layer read is listed in the class in which the code reads
    it: supported knobs (env_int() / getenv, every build) in include/mtgpu.h "Environment"; experiment knobs (exp_int():
    constants unless built with -DMTGPU_EXPERIMENTS) in csrc/knobs.h only — the product header does not carry them."""
    import re
    csrc = os.path.join(os.path.dirname(m.LIB_PATH), "csrc")
    files = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h"))]
    files += [os.path.join(csrc, "host", f) for f in os.listdir(os.path.join(csrc, "host"))]
    supported, experiments = set(), set()
    for path in files:
        text = open(path).read()
        experiments |= set(re.findall(r'exp_int\("(MTGPU_[A-Z0-9_]+)"', text))
        supported |= set(re.findall(r'(?:env_int|env_i|getenv)\("(MTGPU_[A-Z0-9_]+)"', text))
    header = open(os.path.join(ROOT, "include", "mtgpu.h")).read()
    table = header[header.index(" * Environment (read once"):header.index("#ifndef MTGPU_H")]
    knobs = open(os.path.join(csrc, "knobs.h")).read()
    exp_part = knobs[knobs.index("//    experiments   "):knobs.index("#pragma once")]
    sup_part = table
    listed = lambda part: set(re.findall(r"MTGPU_[A-Z0-9_]+", part.replace("MTGPU_INJECT_SUBMIT_FAIL / _GROW_FAIL / _COLLECT_FAIL",
                                                                           "MTGPU_INJECT_SUBMIT_FAIL MTGPU_INJECT_GROW_FAIL MTGPU_INJECT_COLLECT_FAIL")))
    assert experiments and supported and not (experiments & supported - {"MTGPU_FORCE_BLOCK"})
    assert experiments <= listed(exp_part), sorted(experiments - listed(exp_part))
    assert supported <= listed(sup_part) | {"MTGPU_LIBRARY"}, sorted(supported - listed(sup_part))
    # nothing is advertised that no code reads (MTGPU_FORCE_BLOCK appears in both classes: 512 | 1024 always, 256 in the experiments build)
    assert listed(exp_part) - {"MTGPU_FORCE_BLOCK"} <= experiments, sorted(listed(exp_part) - experiments)
    assert listed(sup_part) <= supported, sorted(listed(sup_part) - supported)
