"""Frames made of RUNS of same-cell records (shared by the GPU parity tests of the run-aggregated vote path and by
the CPU-tier oracle-vs-numpy cross-check): what a codec emits for one block — two prediction directions, partitions —
back to back."""
import numpy as np

import mvtrim_amd as m


def run_frames(rng, width, height, shift, vec, n_frames, margin_rows, totals=None, d=3):
    """Frames made of RUNS: consecutive records that land in the same cell (what a codec emits for one block: two
    prediction directions, partitions).  Every frame holds 1-3 pairs of 4-neighbouring cells; a cell's votes
    (vec - 1, vec, vec + 1 or hundreds: the flag hinges on the exact count) arrive as several runs of random length,
    scattered at random positions — hence at every lane alignment and across wave / step boundaries — through a
    stream of records below the threshold."""
    gw, gh = (width + (1 << shift) - 1) >> shift, (height + (1 << shift) - 1) >> shift
    totals = totals or [max(vec - 1, 0), vec, vec + 1, 5 * vec + 64, 700]
    frames = []
    for _ in range(n_frames):
        runs = []                                            # (cx, cy, length)
        for _ in range(int(rng.randint(1, 4))):
            gx, gy = int(rng.randint(1, gw - 2)), int(rng.randint(margin_rows, gh - margin_rows - 1))
            for (cx, cy) in ((gx, gy), (gx + 1, gy) if rng.rand() < 0.5 else (gx, gy + 1)):
                left = int(totals[rng.randint(0, len(totals))])
                while left > 0:
                    n = min(left, int(rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 63, 64, 65, 130])))
                    runs.append((cx, cy, n))
                    left -= n
        order = rng.permutation(len(runs))
        recs = []
        for i in order:
            cx, cy, n = runs[i]
            k = int(rng.choice([0, 0, 1, 2, 3, 5, 17, 60, 64, 200]))
            fx, fy = rng.randint(0, width, size=k), rng.randint(0, height, size=k)
            recs += [(int(a), int(b), 0) for a, b in zip(fx, fy)]
            x = (cx << shift) + rng.randint(0, 1 << shift, size=n)
            y = (cy << shift) + rng.randint(0, 1 << shift, size=n)
            recs += [(int(a), int(b), d) for a, b in zip(x, y)]
        arr = np.array(recs, dtype=np.int64).reshape(-1, 3)
        mv = np.zeros(len(arr), dtype=m.MV_DTYPE)
        mv["dst_x"], mv["dst_y"] = arr[:, 0], arr[:, 1]
        mv["src_x"], mv["src_y"] = arr[:, 0] - arr[:, 2], arr[:, 1]
        frames.append(mv)
    return frames
