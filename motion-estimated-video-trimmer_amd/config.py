"""Environment-variable configuration, mirroring the reference's Config getters
(include/motion_trim/config.hpp:28-125) for the values the scan path consumes.

Same variable names, same code defaults, same parse types (stod / stoi / stof) and
the same uint8 cast for VECTORS_NEEDED (config.hpp:75).  Unlike the reference's
function-local statics the values are read at call time, so tests can vary them.
"""
import os

import numpy as np


def _env(name, default, conv):
    v = os.environ.get(name)
    return conv(v) if v is not None else default


def mv_threshold_sq():      # config.hpp:56-59
    return _env("MV_THRESHOLD_SQ", 16.0, float)


def block_size():           # config.hpp:62-65
    return _env("BLOCK_SIZE", 16, int)


def block_shift():          # config.hpp:68-71
    return _env("BLOCK_SHIFT", 4, int)


def vectors_needed():       # config.hpp:74-77  static_cast<uint8_t>(int)
    return _env("VECTORS_NEEDED", 2, int) & 0xFF


def clusters_needed():      # config.hpp:80-83
    return _env("CLUSTERS_NEEDED", 2, int)


def vertical_mask():        # config.hpp:86-89  (float32)
    return float(np.float32(_env("VERTICAL_MASK", 0.05, float)))


def max_gap_sec():          # config.hpp:92-95
    return _env("MAX_GAP_SEC", 5.0, float)


def padding_sec():          # config.hpp:98-101
    return _env("PADDING_SEC", 0.5, float)


def chunk_duration_sec():   # config.hpp:104-107
    return _env("CHUNK_DURATION_SEC", 30.0, float)


def target_fps():           # config.hpp:113-116
    return _env("TARGET_FPS", 0.0, float)


def min_savings_pct():      # config.hpp:122-125
    return _env("MIN_SAVINGS_PCT", 5.0, float)


# The two parameter sets SURVEY.md §5 documents: code defaults and the shipped env file
# (config/motion_trim.env:36,75,93,113).
CODE_DEFAULTS = dict(mv_threshold_sq=16.0, block_size=16, block_shift=4, vectors_needed=2,
                     clusters_needed=2, vertical_mask=0.05)
SHIPPED_ENV = dict(mv_threshold_sq=4.0, block_size=16, block_shift=4, vectors_needed=4,
                   clusters_needed=2, vertical_mask=0.05)
