"""Soak: randomised GPU-vs-oracle parity for MTGPU_SOAK_SECONDS (default 4 s; set it to minutes
to hunt rare races in the LDS vote / slice hand-off paths).  Every iteration draws a new grid,
parameter set, counter form, slice count and ragged adversarial batch."""
import os
import time

import numpy as np
import pytest

import mvtrim_amd as m
from mvtrim_amd import synth

import oracle_binding as ob
from soak_replay import draw_head, draw_tail

pytestmark = pytest.mark.gpu


def test_soak_random_parity(gpu_scanner_factory):
    budget = float(os.environ.get("MTGPU_SOAK_SECONDS", "4"))
    seed = int(os.environ.get("MTGPU_SOAK_SEED", "12345"))
    rng = np.random.RandomState(seed)
    t_end = time.time() + budget
    it = done = 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog_dir = os.path.join(root, "gpurun_out")
    last_note = time.time()
    while time.time() < t_end:
        it += 1
        if time.time() - last_note > 30 and os.path.isdir(prog_dir):      # keep long runs visibly alive
            last_note = time.time()
            with open(os.path.join(prog_dir, "soak_progress.log"), "a") as fh:
                fh.write(f"{time.strftime('%H:%M:%S')} seed {seed}: {done} configurations ok\n")
        head = draw_head(rng, it)
        w, h, kw, knobs = head["w"], head["h"], head["kw"], head["knobs"]
        p = ob.params_from_config(w, h, **kw)
        for k_, v_ in knobs.items():
            if v_:
                os.environ[k_] = str(v_)
        try:
            s = gpu_scanner_factory(p, force_fb=head["force_fb"])
        except m.MtgpuError as e:
            assert e.code == 2
            continue
        finally:
            for k_ in knobs:
                os.environ.pop(k_, None)
        tail = draw_tail(rng, it, head)
        s.set_slices(tail["slices"])
        n_frames, mv, off, sd = tail["n_frames"], tail["mv"], tail["off"], tail["sd"]
        want = ob.scan_frames(p, mv, off, sd, nthreads=8)
        for _ in range(2):                              # twice: warm caches, reused workspaces
            got = s.check_frames(m.FrameBatch(mv, off, None, sd))
            assert np.array_equal(got, want), (seed, it, w, h, kw, s.plan, knobs)
        if it % 3 == 0:                                 # the 8-byte compact layout, device-resident
            import torch
            rec = m.pack_records(mv)
            d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda() if len(rec) else \
                torch.zeros(8, dtype=torch.uint8, device="cuda")
            got = s.check_frames_device_compact(d_rec[: len(rec) * 8], torch.from_numpy(off.astype(np.int64)).cuda(),
                                                torch.from_numpy(sd).cuda()).cpu().numpy()
            assert np.array_equal(got, want), ("compact", seed, it, w, h, kw, s.plan, knobs)
        if it % 2 == 0:                                 # the pinned pipe (zero-copy compact staging): every other configuration
                                                        # since round 4 — staging reuse is where a visibility fault would show
            pipe = m.ScanPipe(s, *tail["pipe"])
            for f in range(n_frames):
                fr = mv[int(off[f]):int(off[f + 1])]
                pipe.feed(fr if sd[f] else None, float(f), tag=f)
            out = pipe.drain()
            pipe.close()
            assert [fl for _, fl, _ in out] == want.tolist(), ("pipe", seed, it, w, h, kw, s.plan)
        s.close()
        done += 1
    assert done > 0
    print(f"soak: {done} random configurations checked in {budget:.0f} s")
