"""The host copy-out loops (csrc/pack_simd.cpp): every vector loop writes exactly the bytes the scalar
loop writes — bytes 6..13 of each 40-byte AVMotionVector (reference src/motion_scanner.cpp:246-256 reads
nothing else), in order — for every source and destination alignment, 0..130 records, with junk in all the
bytes that must NOT be kept, and never a byte outside the destination range.  CPU tier: data movement
only, no device."""
import ctypes as C

import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import _abi

PF = 16 << 8        # MT_PACK_PREFETCH_LINES(16): software prefetch 1 KiB ahead (runs past the source's end: must be harmless)
LOOPS = [(_abi.PACK_SCALAR, "scalar"), (_abi.PACK_AVX2, "avx2"), (_abi.PACK_AVX2 | _abi.PACK_NT, "avx2_nt"),
         (_abi.PACK_AVX512, "avx512"), (_abi.PACK_AVX512 | _abi.PACK_NT, "avx512_nt"),
         (_abi.PACK_AVX2 | _abi.PACK_NT | PF, "avx2_nt_prefetch"), (_abi.PACK_AVX512 | _abi.PACK_NT | PF, "avx512_nt_prefetch")]


def _want(src: np.ndarray, n: int) -> bytes:
    return src[: n * 40].reshape(n, 40)[:, 6:14].tobytes()


def _run(lib, flags, src_view, n, dst_view):
    return lib.mtgpu_pack_records_with(flags, _addr(src_view), n, _addr(dst_view))


def _addr(a):
    """Address of a view's first byte, also for an empty view (numpy reports a dummy address there)."""
    return a.ctypes.data if a.size else (a.base.ctypes.data if a.base is not None else a.ctypes.data)


@pytest.mark.parametrize("flags,name", LOOPS, ids=[n for _, n in LOOPS])
def test_vector_pack_equals_scalar_pack_every_alignment(flags, name):
    lib = m.load_library()
    rng = np.random.default_rng(1234 + flags)
    probe_src = np.zeros(40 * 16, np.uint8)
    probe_dst = np.zeros(8 * 16 + 64, np.uint8)
    if _run(lib, flags, probe_src, 16, probe_dst) == _abi.MT_ERR_UNSUPPORTED:
        pytest.skip(f"this CPU cannot run the {name} loop")
    guard = 0xC5
    for n in list(range(0, 131)) + [255, 256, 257, 1000, 4093]:
        src_buf = rng.integers(0, 256, size=n * 40 + 64 + 64, dtype=np.uint8)      # junk everywhere
        for sa in (0, 1, 2, 6, 8, 16, 24, 40 % 64, 63):                           # source alignment mod 64
            src = src_buf[sa: sa + n * 40]
            want = _want(src, n)
            for da in (0, 8, 16, 24, 32, 40, 48, 56, 1, 4, 62):                    # destination alignment mod 64
                dst_buf = np.full(n * 8 + 256, guard, np.uint8)
                base = (-dst_buf.ctypes.data) % 64 + 64 + da                       # 64-aligned + da, 64+ guard bytes before
                dst = dst_buf[base: base + n * 8]
                assert (dst_buf.ctypes.data + base - da) % 64 == 0
                assert _run(lib, flags, src, n, dst) == _abi.MT_OK
                assert dst.tobytes() == want, (name, n, sa, da)
                assert (dst_buf[:base] == guard).all() and (dst_buf[base + n * 8:] == guard).all(), (name, n, sa, da)


def test_selected_loop_is_one_this_cpu_runs_and_matches_scalar():
    lib = m.load_library()
    sel = lib.mtgpu_pack_selected()
    assert (sel & _abi.PACK_IMPL_MASK) in (_abi.PACK_SCALAR, _abi.PACK_AVX2, _abi.PACK_AVX512)
    rng = np.random.default_rng(7)
    mv = rng.integers(0, 256, size=40 * 777, dtype=np.uint8).view(m.MV_DTYPE)
    got = m.pack_records(mv)                                                        # the auto-dispatched loop
    ref = np.zeros(777, m.COMPACT_DTYPE)
    assert lib.mtgpu_pack_records_with(_abi.PACK_SCALAR, mv.ctypes.data, 777, ref.ctypes.data) == _abi.MT_OK
    assert got.tobytes() == ref.tobytes() == _want(mv.view(np.uint8), 777)
    # the loop the dispatcher chose is runnable by definition
    out = np.zeros(777, m.COMPACT_DTYPE)
    assert lib.mtgpu_pack_records_with(sel, mv.ctypes.data, 777, out.ctypes.data) == _abi.MT_OK
    assert out.tobytes() == ref.tobytes()


def test_pack_with_rejects_bad_arguments():
    lib = m.load_library()
    buf = np.zeros(80, np.uint8)
    assert lib.mtgpu_pack_records_with(0, buf.ctypes.data, 1, buf.ctypes.data) == _abi.MT_ERR_INVALID
    assert lib.mtgpu_pack_records_with(4, buf.ctypes.data, 1, buf.ctypes.data) == _abi.MT_ERR_INVALID
    assert lib.mtgpu_pack_records_with(_abi.PACK_SCALAR | (1 << 16), buf.ctypes.data, 1, buf.ctypes.data) == _abi.MT_ERR_INVALID
    assert lib.mtgpu_pack_records_with(_abi.PACK_SCALAR, None, 1, buf.ctypes.data) == _abi.MT_ERR_INVALID
    assert lib.mtgpu_pack_records_with(_abi.PACK_SCALAR, None, 0, None) == _abi.MT_OK


def test_env_override_pins_the_loop():
    """MTGPU_PACK / MTGPU_PACK_NT are read once per process: checked in a child."""
    import subprocess, sys, os
    code = ("import mvtrim_amd as m; lib = m.load_library(); print(lib.mtgpu_pack_selected())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def sel(**env):
        e = dict(os.environ, **env)
        return int(subprocess.check_output([sys.executable, "-c", code], cwd=root, env=e).decode().split()[-1])
    assert sel(MTGPU_PACK="scalar") == _abi.PACK_SCALAR
    auto = sel()
    if (auto & _abi.PACK_IMPL_MASK) != _abi.PACK_SCALAR:
        assert auto & _abi.PACK_NT
        # MTGPU_PACK_NT is an A/B knob (csrc/knobs.h): the default build ignores it
        exp = b"+experiments" in m.load_library().mtgpu_version()
        assert sel(MTGPU_PACK_NT="0") == ((auto & _abi.PACK_IMPL_MASK) if exp else auto)
    assert sel(MTGPU_PACK="nonsense") == auto
