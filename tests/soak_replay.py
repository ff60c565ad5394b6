"""The random stream of tests/test_gpu_soak.py, in two halves around the scanner creation, so that the soak and a
CPU-side replay (scripts/replay_soak_config.py, the regression test of round 4's one wrong flag) draw the same numbers.

    head = draw_head(rng, it)                  # grid, parameters, knobs, forced counter form
    ... create the scanner (the soak) / preview its plan (the replay); MT_ERR_CAPACITY -> next iteration ...
    tail = draw_tail(rng, it, head, pipe_every)   # slices, frames, records, runs, pipe geometry

`pipe_every`: the soak fed the pipe every 5th iteration when that flag was recorded, every 2nd since 5145aa9."""
import numpy as np

from mvtrim_amd import synth

FORMS = [None, None, 1, 2, 4, 8, 108, 32]


def draw_head(rng, it):
    sh = int(rng.randint(1 if it % 7 == 0 else 2, 6))        # shift 1: grids up to ~1900x1100 -> row bands
    w, h = int(rng.randint(64, 3900)), int(rng.randint(64, 2200))
    kw = dict(mv_threshold_sq=float(rng.choice([16.0, 4.0, 0.0, 9.5])), block_size=1 << sh, block_shift=sh,
              vectors_needed=int(rng.choice([1, 1, 2, 2, 3, 4, 6, 12, 255])),
              clusters_needed=int(rng.choice([1, 2, 2, 3, 10])),
              vertical_mask=float(rng.choice([0.0, 0.05, 0.2])))
    # frames per workgroup (next-frame prefetch on compact records), line alignment: knobs read at create time
    # (MTGPU_PREFETCH / MTGPU_ALIGN are honoured by the experiments build only)
    knobs = {"MTGPU_GROUP": str(rng.choice(["", "", "2", "3", "8"])), "MTGPU_PREFETCH": str(rng.choice(["", "", "0"])),
             "MTGPU_ALIGN": str(rng.choice(["", "", "0"]))}
    return dict(w=w, h=h, kw=kw, knobs=knobs, force_fb=FORMS[it % len(FORMS)])


def draw_tail(rng, it, head, pipe_every=2):
    w, h = head["w"], head["h"]
    slices = int(rng.choice([0, 1, 2, 4, 8]))
    n_frames = int(rng.choice([3, 17, 64, 300]))
    mv, off, sd = synth.random_frames(rng, n_frames, int(rng.choice([200, 3000, 20000, 20000 if n_frames > 64 else 60000])), w, h,
                                      hot=float(rng.choice([0.05, 0.5, 0.95])))
    if it % 2 == 0 and len(mv):                   # runs: every record repeated 1..6 times back to back (a block's
        r = rng.randint(1, 7, size=len(mv))         # several vectors) — the run-aggregated vote path of packed forms
        csum = np.concatenate([[0], np.cumsum(r)])
        off = csum[off.astype(np.int64)].astype(np.uint64)
        mv = np.repeat(mv, r)
    pipe = None
    if it % pipe_every == 0:
        pipe = (int(rng.choice([500, 5000, 50000])), int(rng.choice([1, 4, 32])), int(rng.choice([1, 2, 3])))
    return dict(slices=slices, n_frames=n_frames, mv=mv, off=off, sd=sd, pipe=pipe)


def replay(seed, target, pipe_every=2):
    """Iteration `target` of the soak with `seed`, reconstructed without a GPU: (head, tail, plan)."""
    import os
    import mvtrim_amd as m
    import oracle_binding as ob
    rng = np.random.RandomState(seed)
    it = 0
    while True:
        it += 1
        head = draw_head(rng, it)
        p = ob.params_from_config(head["w"], head["h"], **head["kw"])
        old = os.environ.get("MTGPU_FORCE_FB")
        try:
            if head["force_fb"] is None:
                os.environ.pop("MTGPU_FORCE_FB", None)
            else:
                os.environ["MTGPU_FORCE_FB"] = str(head["force_fb"])
            plan = m.plan_preview(p)                  # MT_ERR_CAPACITY here = mtgpu_create refusing the plan
        except m.MtgpuError as e:
            assert e.code == 2
            continue
        finally:
            if old is None:
                os.environ.pop("MTGPU_FORCE_FB", None)
            else:
                os.environ["MTGPU_FORCE_FB"] = old
        tail = draw_tail(rng, it, head, pipe_every)
        if it == target:
            return head, tail, plan, p


def pipe_batches(off, sd, max_records, max_frames, n_buffers):
    """The batching ScanPipe.feed() performs on these frames (mtgpu_batch_add_frame: MT_ERR_CAPACITY -> submit, next
    batch; a frame larger than a whole EMPTY batch grows that batch's staging to n + n / 4 records; batches rotate
    through the staging blocks in acquire order).  Returns ([(staging block, [frames], records)], [(frame, block, n)])."""
    caps = [max_records] * n_buffers
    batches, grows, cur, cur_rec, b = [], [], [], 0, 0
    for f in range(len(sd)):
        n = int(off[f + 1] - off[f]) if sd[f] else 0
        if cur and (len(cur) >= max_frames or cur_rec + n > caps[b % n_buffers]):
            batches.append((b % n_buffers, cur, cur_rec))
            b += 1
            cur, cur_rec = [], 0
        if not cur and n > caps[b % n_buffers]:
            caps[b % n_buffers] = n + n // 4
            grows.append((f, b % n_buffers, n))
        cur.append(f)
        cur_rec += n
    if cur:
        batches.append((b % n_buffers, cur, cur_rec))
    return batches, grows
