"""Replays the random stream of tests/test_gpu_soak.py on the CPU (no GPU call is made) up to one iteration and
prints what that iteration fed to the pipe: pipe geometry, which frames opened a batch, which frames forced the
staging to grow.  Used to read the one wrong flag of round 4 (profiles/r04_soak_mismatch_with_registered_staging.txt:
seed 10242, iteration 70, frame 33) from the evidence in hand instead of re-running the soak.

usage: python scripts/replay_soak_config.py SEED ITERATION [FRAME [PIPE_EVERY]]
(PIPE_EVERY: the soak fed the pipe every 5th iteration when the mismatch was recorded, every 2nd since 5145aa9)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mvtrim_amd as m            # noqa: E402
from mvtrim_amd import synth      # noqa: E402
import oracle_binding as ob       # noqa: E402


def main():
    seed, target = int(sys.argv[1]), int(sys.argv[2])
    frame = int(sys.argv[3]) if len(sys.argv) > 3 else None
    pipe_every = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    rng = np.random.RandomState(seed)
    forms = [None, None, 1, 2, 4, 8, 108, 32]
    it = 0
    while True:
        it += 1
        sh = int(rng.randint(1 if it % 7 == 0 else 2, 6))
        w, h = int(rng.randint(64, 3900)), int(rng.randint(64, 2200))
        kw = dict(mv_threshold_sq=float(rng.choice([16.0, 4.0, 0.0, 9.5])), block_size=1 << sh, block_shift=sh,
                  vectors_needed=int(rng.choice([1, 1, 2, 2, 3, 4, 6, 12, 255])),
                  clusters_needed=int(rng.choice([1, 2, 2, 3, 10])),
                  vertical_mask=float(rng.choice([0.0, 0.05, 0.2])))
        p = ob.params_from_config(w, h, **kw)
        knobs = {"MTGPU_GROUP": rng.choice(["", "", "2", "3", "8"]), "MTGPU_PREFETCH": rng.choice(["", "", "0"]),
                 "MTGPU_ALIGN": rng.choice(["", "", "0"])}
        force = forms[it % len(forms)]
        if force is None:
            os.environ.pop("MTGPU_FORCE_FB", None)
        else:
            os.environ["MTGPU_FORCE_FB"] = str(force)
        try:
            plan = m.plan_preview(p)                      # MT_ERR_CAPACITY here = mtgpu_create refusing the plan
        except m.MtgpuError as e:
            assert e.code == 2
            continue
        finally:
            os.environ.pop("MTGPU_FORCE_FB", None)
        slices = int(rng.choice([0, 1, 2, 4, 8]))
        n_frames = int(rng.choice([3, 17, 64, 300]))
        mv, off, sd = synth.random_frames(rng, n_frames, int(rng.choice([200, 3000, 20000, 20000 if n_frames > 64 else 60000])),
                                          w, h, hot=float(rng.choice([0.05, 0.5, 0.95])))
        if it % 2 == 0 and len(mv):
            r = rng.randint(1, 7, size=len(mv))
            csum = np.concatenate([[0], np.cumsum(r)])
            off = csum[off.astype(np.int64)].astype(np.uint64)
            mv = np.repeat(mv, r)
        pipe = None
        if it % pipe_every == 0:
            pipe = (int(rng.choice([500, 5000, 50000])), int(rng.choice([1, 4, 32])), int(rng.choice([1, 2, 3])))
        if it == target:
            break
    want = ob.scan_frames(p, mv, off, sd, nthreads=8)
    print(f"seed {seed} iteration {it}: {w}x{h} {kw}")
    print(f"  plan {plan}  forced form {force}  slices request {slices}  knobs {knobs}")
    print(f"  frames {n_frames}, records {len(mv)}, frames with side data {int(sd.sum())}, flags set {int(want.sum())}")
    if pipe is None:
        print("  no pipe in this iteration")
        return
    max_rec, max_fr, nbuf = pipe
    print(f"  pipe: max_records {max_rec}, max_frames {max_fr}, n_buffers {nbuf}")
    # the batching ScanPipe.feed() performs (mtgpu_batch_add_frame: MT_ERR_CAPACITY -> submit, next batch;
    # a frame larger than a whole EMPTY batch grows that batch's staging to n + n / 4 records)
    caps = [max_rec] * nbuf                               # per staging batch (they rotate in acquire order)
    batches, cur, cur_rec, b = [], [], 0, 0
    grows = []
    for f in range(n_frames):
        n = int(off[f + 1] - off[f]) if sd[f] else 0
        if cur and (len(cur) >= max_fr or cur_rec + n > caps[b % nbuf]):
            batches.append((b % nbuf, cur, cur_rec))
            b += 1
            cur, cur_rec = [], 0
        if not cur and n > caps[b % nbuf]:
            caps[b % nbuf] = n + n // 4
            grows.append((f, b % nbuf, n))
        cur.append(f)
        cur_rec += n
    if cur:
        batches.append((b % nbuf, cur, cur_rec))
    print(f"  {len(batches)} batches over {nbuf} staging block(s); {len(grows)} staging re-pins (oversize frames): "
          f"{grows[:12]}{' ...' if len(grows) > 12 else ''}")
    if frame is not None:
        for i, (slot, fr, nrec) in enumerate(batches):
            if frame in fr:
                print(f"  frame {frame}: batch {i} (staging block {slot}), position {fr.index(frame)} of {len(fr)}, "
                      f"batch records {nrec}, frame records {int(off[frame + 1] - off[frame])}, oracle flag {int(want[frame])}, "
                      f"grown for this frame: {any(g[0] == frame for g in grows)}")
                prev = [x for x in batches[:i] if x[0] == slot]
                print(f"  earlier batches through the same staging block: {len(prev)}"
                      + (f"; the one before held frames {prev[-1][1][0]}..{prev[-1][1][-1]}, its flag at the same position: "
                         f"{int(want[prev[-1][1][fr.index(frame)]]) if fr.index(frame) < len(prev[-1][1]) else 'none'}" if prev else ""))


if __name__ == "__main__":
    main()
