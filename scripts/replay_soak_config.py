"""Replays the random stream of tests/test_gpu_soak.py on the CPU (no GPU call is made) up to one iteration and
prints what that iteration fed to the pipe: pipe geometry, which frames opened a batch, which frames forced the
staging to grow.  Written to read the one wrong flag of round 4 (profiles/r04_soak_mismatch_with_registered_staging.txt:
seed 10242, iteration 70, frame 33) from the evidence in hand instead of re-running the soak.

usage: python scripts/replay_soak_config.py SEED ITERATION [FRAME [PIPE_EVERY]]
(PIPE_EVERY: the soak fed the pipe every 5th iteration when the mismatch was recorded, every 2nd since 5145aa9)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob       # noqa: E402
from soak_replay import pipe_batches, replay   # noqa: E402


def main():
    seed, target = int(sys.argv[1]), int(sys.argv[2])
    frame = int(sys.argv[3]) if len(sys.argv) > 3 else None
    pipe_every = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    head, tail, plan, p = replay(seed, target, pipe_every)
    mv, off, sd, n_frames = tail["mv"], tail["off"], tail["sd"], tail["n_frames"]
    want = ob.scan_frames(p, mv, off, sd, nthreads=8)
    print(f"seed {seed} iteration {target}: {head['w']}x{head['h']} {head['kw']}")
    print(f"  plan {plan}  forced form {head['force_fb']}  slices request {tail['slices']}  knobs {head['knobs']}")
    print(f"  frames {n_frames}, records {len(mv)}, frames with side data {int(sd.sum())}, flags set {int(want.sum())}")
    if tail["pipe"] is None:
        print("  no pipe in this iteration")
        return
    max_rec, max_fr, nbuf = tail["pipe"]
    print(f"  pipe: max_records {max_rec}, max_frames {max_fr}, n_buffers {nbuf}")
    batches, grows = pipe_batches(off, sd, *tail["pipe"])
    print(f"  {len(batches)} batches over {nbuf} staging block(s); {len(grows)} staging re-pins (oversize frames): "
          f"{grows[:12]}{' ...' if len(grows) > 12 else ''}")
    if frame is not None:
        for i, (slot, fr, nrec) in enumerate(batches):
            if frame in fr:
                print(f"  frame {frame}: batch {i} (staging block {slot}), position {fr.index(frame)} of {len(fr)}, "
                      f"batch records {nrec}, frame records {int(off[frame + 1] - off[frame])}, oracle flag {int(want[frame])}, "
                      f"grown for this frame: {any(g[0] == frame for g in grows)}")
                prev = [x for x in batches[:i] if x[0] == slot]
                print(f"  earlier batches through the same staging block: {len(prev)}")


if __name__ == "__main__":
    main()
