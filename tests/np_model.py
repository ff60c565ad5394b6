"""A second, independent restatement of check_frame in vectorised numpy (TEST ONLY).

Written differently from oracle/mt_oracle.c on purpose (histogram + shifted boolean
planes instead of a scalar loop with early exit) so that the two restatements can be
checked against each other on random inputs.  Reference: src/motion_scanner.cpp:217-295.
"""
import numpy as np


def check_frame_np(p, mv, has_sd=True):
    """p: mvtrim_amd.ScanParams.  Returns (flag, centres)."""
    if not has_sd:
        return 0, 0
    gw, gh, m = p.grid_w, p.grid_h, p.vertical_margin
    votes = np.zeros((gh, gw), dtype=np.int64)
    if len(mv):
        dx = mv["dst_x"].astype(np.int64) - mv["src_x"].astype(np.int64)
        dy = mv["dst_y"].astype(np.int64) - mv["src_y"].astype(np.int64)
        mag = dx * dx + dy * dy
        keep = ~(mag.astype(np.float64) < p.mv_threshold_sq)        # NaN threshold keeps all
        gx = mv["dst_x"].astype(np.int64) >> p.block_shift          # arithmetic shift
        gy = mv["dst_y"].astype(np.int64) >> p.block_shift
        keep &= (gx >= 0) & (gx < gw) & (gy >= m) & (gy < gh - m)
        np.add.at(votes, (gy[keep], gx[keep]), 1)
    votes = np.minimum(votes, 255)                                  # u8 saturation
    act = votes >= p.vectors_needed
    nb = np.zeros_like(act)
    nb[:, 1:] |= act[:, :-1]
    nb[:, :-1] |= act[:, 1:]
    nb[1:, :] |= act[:-1, :]
    nb[:-1, :] |= act[1:, :]
    centre = act & nb
    centre[:, 0] = False
    centre[:, gw - 1:] = False
    rows = np.zeros(gh, dtype=bool)
    rows[max(m, 0):max(gh - m, 0)] = True
    centre &= rows[:, None]
    n = int(centre.sum())
    return int(n >= max(1, p.clusters_needed)), n
