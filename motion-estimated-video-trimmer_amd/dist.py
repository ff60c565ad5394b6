"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

check_frame is stateless across frames (the vote grid is zeroed per frame,
src/motion_scanner.cpp:229), so the scan shards with NO data-path collective.
The path has exactly one exchange step, after the scan:

  * stream sharding (BASELINE.json config 4: 64 streams over 8 GPUs): every rank owns
    whole streams, merges them locally on its GPU, and the fixed-capacity per-stream
    segment lists are all-gathered once  -> `gather_segment_lists`;
  * time-range sharding of ONE stream (the reference's chunk parallelism,
    src/pipeline.cpp:163-167, spread over GPUs): every rank scans a contiguous frame
    range and contributes its compacted motion timestamps; they are all-gathered and
    merged once, which is bit-identical to the single-device merge because the merge
    sorts and de-duplicates its input (src/pipeline.cpp:302-304) -> `gather_timestamps`.

Messages are tens of 16-byte segments per stream (<< 1 MB): the collective is
latency-bound, so one packed all_gather per batch is the whole design; bucket
sizes / ring-vs-tree are immaterial at this size.
"""
from typing import List, Sequence, Tuple

import numpy as np

from ._abi import MERGE_RESULT_DTYPE


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of n_items for `rank` (first n%world ranks get +1)."""
    q, r = divmod(n_items, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


def shard_by_records(frame_off: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous frame ranges with ~equal RECORD counts (the scan is bandwidth bound, so
    bytes — not frames — are what must balance).  frame_off: CSR offsets [F+1]."""
    off = np.asarray(frame_off, dtype=np.int64)
    n = len(off) - 1
    total = int(off[-1] - off[0])
    cuts = [0]
    for r in range(1, world):
        target = off[0] + (total * r) // world
        f = int(np.searchsorted(off, target, side="left"))
        cuts.append(min(max(f, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def pack_segment_lists(seg, res):
    """[S, cap, 2] float64 segments + [S, 40] uint8 results -> one uint8 [S, cap*16 + 40]."""
    import torch
    s = seg.shape[0]
    return torch.cat([seg.reshape(s, -1).contiguous().view(torch.uint8), res], dim=1).contiguous()


def unpack_segment_lists(packed, cap: int):
    """Inverse of pack_segment_lists for a [..., S, cap*16+40] uint8 tensor (any device):
    returns (segments float64 [..., S, cap, 2] as numpy, results structured numpy [..., S])."""
    a = packed.cpu().numpy()
    lead = a.shape[:-1]
    seg = np.ascontiguousarray(a[..., : cap * 16]).view(np.float64).reshape(lead + (cap, 2))
    res = np.ascontiguousarray(a[..., cap * 16:]).view(MERGE_RESULT_DTYPE).reshape(lead)
    return seg, res


def gather_segment_lists(seg, res, group=None, out=None, s_pad=None):
    """All-gather the per-rank segment lists (stream sharding).  seg: [S, cap, 2] float64,
    res: [S, 40] uint8; cap is the same on every rank, S may differ by rank when `s_pad`
    (>= every rank's S, e.g. ceil(n_streams / world)) is given: shorter ranks are zero-padded.
    Returns uint8 [world, s_pad or S, cap*16+40] on the input device; decode with
    unpack_segment_lists / assemble_stream_lists."""
    import torch
    import torch.distributed as dist
    packed = pack_segment_lists(seg, res)
    if s_pad is not None and packed.shape[0] < s_pad:
        fill = torch.zeros((s_pad - packed.shape[0], packed.shape[1]), dtype=torch.uint8, device=packed.device)
        packed = torch.cat([packed, fill], dim=0)
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world,) + tuple(packed.shape), dtype=torch.uint8, device=packed.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, packed, group=group)
    else:
        parts = [out[i] for i in range(world)]
        dist.all_gather(parts, packed, group=group)
    return out


def gather_timestamps(ts, group=None):
    """All-gather variable-length per-rank motion timestamp lists (time-range sharding of
    one stream).  ts: 1-D float64 tensor (this rank's compacted timestamps, any length).
    Returns the pooled 1-D tensor, rank-major.  Two small collectives: counts, then the
    payload padded to the maximum count."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([ts.numel()], dtype=torch.int64, device=ts.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    mine = torch.zeros(cap, dtype=torch.float64, device=ts.device)
    mine[: ts.numel()] = ts
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def assemble_stream_lists(gathered, cap: int, streams_per_rank: Sequence[int]) -> List[dict]:
    """Decode a gather_segment_lists result into one entry per global stream, rank-major:
    {"rank", "local_stream", "segments" (k,2) float64, "result" mt_merge_result record}."""
    seg, res = unpack_segment_lists(gathered, cap)
    out = []
    for r in range(seg.shape[0]):
        for s in range(streams_per_rank[r]):
            k = min(int(res[r, s]["n_segments"]), cap)
            out.append({"rank": r, "local_stream": s, "segments": seg[r, s, :k].copy(),
                        "result": res[r, s].copy()})
    return out


def scan_and_merge_timerange(scanner, mv, frame_off, pts, merge_params, has_sd=None, group=None,
                             job_semantics=True, seg_cap=64):
    """Time-range sharding of ONE stream, end to end on the devices: this rank scans its own
    contiguous frame range (device tensors mv / frame_off / pts of that range), the compacted
    motion timestamps of all ranks are all-gathered, and every rank merges the pooled list once
    on its GPU.  Bit-identical to scanning the whole stream on one device, because the merge
    sorts and de-duplicates its input (src/pipeline.cpp:302-304).
    merge_params: MergeParams of the WHOLE stream.  Returns (segments [k,2] numpy, result record)."""
    import torch
    from .scanner import results_from_bytes
    flags = scanner.check_frames_device(mv, frame_off, has_sd)
    local_ts = pts[flags.bool()]
    pooled = gather_timestamps(local_ts, group=group).contiguous()
    dev = pts.device
    seg, res = scanner.merge_timestamps_device(pooled, merge_params, job_semantics, seg_cap)
    torch.cuda.synchronize(dev)
    rec = results_from_bytes(res.cpu().numpy())[0]
    k = min(int(rec["n_segments"]), seg_cap)
    return seg[:k].cpu().numpy(), rec
