"""Environment-variable configuration, mirroring the reference's Config getters
(include/motion_trim/config.hpp:28-125) for the values the scan path consumes.

Same variable names, same code defaults, same parse semantics and the same uint8 cast for
VECTORS_NEEDED (config.hpp:75).  The reference parses with std::stod / std::stoi / std::stof
(config.hpp:28-53), i.e. glibc strtod / strtol / strtof on the longest valid PREFIX ("7abc" -> 7,
"0x10" -> 16.0 as a double but 0 as an int, leading whitespace skipped), std::invalid_argument when
nothing converts and std::out_of_range on ERANGE (overflow AND underflow) or an int outside 32 bits.
The getters below call the same libc functions, so every string gives the reference's value bit for
bit (a float read through Python's float() would be rounded twice); the error cases raise ValueError
(invalid_argument) and OverflowError (out_of_range).  Pinned by tests/golden/reference_host_vectors.json,
which holds the answers of the reference's own config.hpp compiled and run (tests/test_reference_host.py).
Like the reference's function-local statics (config.hpp:56-59) every getter reads and parses its variable
ONCE per process: a later change of the environment does not change the answer, and a value that failed to
parse (the exception leaves the C++ static uninitialised) is parsed again on the next call.  `forget()`
drops the memo (tests only; the reference has no such thing).
"""
import ctypes
import errno
import os

_libc = ctypes.CDLL(None, use_errno=True)
_libc.strtod.restype = ctypes.c_double
_libc.strtof.restype = ctypes.c_float
_libc.strtol.restype = ctypes.c_long
_libc.strtod.argtypes = _libc.strtof.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
_libc.strtol.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]


def _strto(fn, text, what, *base):
    buf = ctypes.create_string_buffer(os.fsencode(text))
    end = ctypes.c_void_p()
    ctypes.set_errno(0)
    v = fn(buf, ctypes.byref(end), *base)
    if (end.value or 0) == ctypes.addressof(buf):
        raise ValueError(f"{what}: no conversion of {text!r}")           # std::invalid_argument
    if ctypes.get_errno() == errno.ERANGE:
        raise OverflowError(f"{what}: {text!r} out of range")            # std::out_of_range
    return v


def stod(text):
    return _strto(_libc.strtod, text, "stod")


def stof(text):
    return _strto(_libc.strtof, text, "stof")


def stoi(text):
    v = _strto(_libc.strtol, text, "stoi", 10)
    if not -2**31 <= v < 2**31:
        raise OverflowError(f"stoi: {text!r} out of range")
    return v


_memo = {}


def forget():
    """Tests only: read the environment afresh on the next call of each getter."""
    _memo.clear()


def _env(name, default, conv):
    if name not in _memo:                       # an exception from conv leaves nothing behind: retried next call
        v = os.environ.get(name)
        _memo[name] = conv(v) if v is not None else default
    return _memo[name]


_F32_0_05 = ctypes.c_float(0.05).value     # the literal 0.05f of config.hpp:87


def mv_threshold_sq():      # config.hpp:56-59
    return _env("MV_THRESHOLD_SQ", 16.0, stod)


def block_size():           # config.hpp:62-65
    return _env("BLOCK_SIZE", 16, stoi)


def block_shift():          # config.hpp:68-71
    return _env("BLOCK_SHIFT", 4, stoi)


def vectors_needed():       # config.hpp:74-77  static_cast<uint8_t>(int)
    return _env("VECTORS_NEEDED", 2, lambda t: stoi(t) & 0xFF)


def clusters_needed():      # config.hpp:80-83
    return _env("CLUSTERS_NEEDED", 2, stoi)


def vertical_mask():        # config.hpp:86-89  (float32)
    return _env("VERTICAL_MASK", _F32_0_05, stof)


def max_gap_sec():          # config.hpp:92-95
    return _env("MAX_GAP_SEC", 5.0, stod)


def padding_sec():          # config.hpp:98-101
    return _env("PADDING_SEC", 0.5, stod)


def chunk_duration_sec():   # config.hpp:104-107
    return _env("CHUNK_DURATION_SEC", 30.0, stod)


def target_fps():           # config.hpp:113-116
    return _env("TARGET_FPS", 0.0, stod)


def min_savings_pct():      # config.hpp:122-125
    return _env("MIN_SAVINGS_PCT", 5.0, stod)


def parallel_streams():     # config.hpp:138-141
    return _env("PARALLEL_STREAMS", 0, stoi)


def threads_per_stream():   # config.hpp:165-168
    return _env("THREADS_PER_STREAM", 0, stoi)


# The two parameter sets SURVEY.md §5 documents: code defaults and the shipped env file
# (config/motion_trim.env:36,75,93,113).
CODE_DEFAULTS = dict(mv_threshold_sq=16.0, block_size=16, block_shift=4, vectors_needed=2,
                     clusters_needed=2, vertical_mask=0.05)
SHIPPED_ENV = dict(mv_threshold_sq=4.0, block_size=16, block_shift=4, vectors_needed=4,
                   clusters_needed=2, vertical_mask=0.05)
