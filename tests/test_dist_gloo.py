"""The N > 1 path on CPU: world_size-2 `gloo` process group, one process per rank.
Covers the exchange step of the path (the only collective): packing, all_gather and
assembly of per-rank segment lists (stream sharding) and of per-rank motion timestamps
(time-range sharding of one stream).  The per-rank scan/merge results are produced by the
oracle here (there is no GPU in this tier); on the GPU box the same dist functions carry
tensors produced by the HIP kernels (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import mvtrim_amd as m
from mvtrim_amd import dist as mdist
from mvtrim_amd import synth

import oracle_binding as ob

CAP = 16


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def stream_case(global_stream):
    """Deterministic per-stream inputs: (flags, pts, merge params)."""
    rng = np.random.RandomState(100 + global_stream)
    n = 200 + 37 * global_stream
    flags = np.zeros(n, dtype=np.uint8)
    for _ in range(1 + global_stream % 3):
        a = rng.randint(0, n - 20)
        flags[a:a + rng.randint(1, 40)] = 1
    if global_stream == 2:
        flags[:] = 0                                   # a stream without motion
    pts = np.arange(n) / 30.0
    mp_ = m.MergeParams(duration=n / 30.0, max_gap_sec=1.0, padding_sec=0.5, min_savings_pct=5.0)
    return flags, pts, mp_


def local_segment_lists(streams):
    seg = np.zeros((len(streams), CAP, 2), dtype=np.float64)
    res = np.zeros(len(streams), dtype=m.MERGE_RESULT_DTYPE)
    for i, g in enumerate(streams):
        flags, pts, mp_ = stream_case(g)
        s, r = ob.pool_and_merge(pts[flags != 0], mp_, True)
        seg[i, :len(s), 0], seg[i, :len(s), 1] = s["start"], s["end"]
        for k, v in r.items():
            res[i][k] = v
    return torch.from_numpy(seg), torch.from_numpy(res.view(np.uint8).reshape(len(streams), -1).copy())


def worker(rank, world, port, n_streams, tmpdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ---- stream sharding: whole streams per rank, one packed all_gather
        b, e = mdist.shard_range(n_streams, world, rank)
        seg, res = local_segment_lists(list(range(b, e)))
        s_pad = (n_streams + world - 1) // world
        gathered = mdist.gather_segment_lists(seg, res, s_pad=s_pad)
        assert tuple(gathered.shape) == (world, s_pad, CAP * 16 + 40)
        per_rank = [mdist.shard_range(n_streams, world, r) for r in range(world)]
        lists = mdist.assemble_stream_lists(gathered, CAP, [y - x for x, y in per_rank])
        assert len(lists) == n_streams
        for g, entry in enumerate(lists):               # every rank holds every stream's list
            flags, pts, mp_ = stream_case(g)
            want_seg, want_res = ob.pool_and_merge(pts[flags != 0], mp_, True)
            assert entry["segments"].tobytes() == np.stack([want_seg["start"], want_seg["end"]], 1).tobytes()
            assert int(entry["result"]["do_cut"]) == want_res["do_cut"]
            assert float(entry["result"]["time_removed"]) == want_res["time_removed"]

        # ---- time-range sharding of ONE stream: gather timestamps, merge once
        spec = synth.spec_1080p(seed=77, sub=1)
        n = 240
        spec.events = [synth.Event(20, 70, 30, 20, 4, 3, 9, 1), synth.Event(100, 130, 60, 40, 3, 3, -8, 2),
                       synth.Event(118, 200, 10, 30, 5, 2, 7, 0)]     # the 2nd/3rd straddle the rank seam
        mv, off, pts, sd = synth.gen_stream(spec, n)
        p = ob.params_from_config(1920, 1080, vectors_needed=1)
        fb, fe = mdist.shard_by_records(off, world)[rank]
        local_off = off[fb:fe + 1]
        flags = ob.scan_frames(p, mv, local_off, sd[fb:fe])          # oracle stands in for the GPU scan
        local_ts = torch.from_numpy(pts[fb:fe][flags != 0])
        pooled = mdist.gather_timestamps(local_ts).numpy()
        mp_ = m.MergeParams(duration=n / 30.0, max_gap_sec=0.5, padding_sec=0.25, min_savings_pct=5.0)
        got_seg, got_res = ob.pool_and_merge(pooled, mp_, True)
        full_flags = ob.scan_frames(p, mv, off, sd)
        want_seg, want_res = ob.pool_and_merge(pts[full_flags != 0], mp_, True)
        assert got_seg.tobytes() == want_seg.tobytes() and got_res == want_res
        assert len(want_seg) >= 2
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_streams", [5, 8])
def test_world2_gloo_gather(tmp_path, n_streams):
    world = 2
    mp.spawn(worker, args=(world, free_port(), n_streams, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
