"""Frame-per-GPU batch driver: shards a batch of independent frames across ranks (one
process per GPU) and gathers the fixed-size per-frame pose records with ONE collective per
batch (torch.distributed all_gather: RCCL over xGMI on GPUs, gloo in the CPU tests).

Frames share nothing - the reference even re-seeds RANSAC per frame, and its only cross-frame
state is the ICP_SUCCESS display latch (iterative_closest_point.cpp:139-147) - so there is
no data-path collective; the gather moves ~1.8 KB per frame and is latency-bound.
"""
import numpy as np

from . import capi


def shard_range(n_frames, rank, world):
    """Contiguous slice [lo, hi) of a batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(local_records, n_frames_total, dist=None, device=None):
    """All-gather per-frame records.  local_records: uint8 array (F_local, FRAME_RESULT_BYTES).
    Returns uint8 array (n_frames_total, FRAME_RESULT_BYTES) on every rank, frame order."""
    import torch
    rec = np.ascontiguousarray(local_records, dtype=np.uint8)
    if dist is None or not dist.is_initialized():
        assert rec.shape[0] == n_frames_total
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    per = max(shard_range(n_frames_total, r, world)[1] - shard_range(n_frames_total, r, world)[0] for r in range(world))
    pad = np.zeros((per, capi.FRAME_RESULT_BYTES), np.uint8)
    pad[:rec.shape[0]] = rec
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device, non_blocking=False)
    out = torch.empty((world * per, capi.FRAME_RESULT_BYTES), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t)
    out = out.cpu().numpy().reshape(world, per, capi.FRAME_RESULT_BYTES)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_frames_total, r, world)
        parts.append(out[r, :hi - lo])
    return np.concatenate(parts, axis=0)


class ShardedBatchRunner:
    """process(frames_of_this_rank) -> records of the WHOLE batch (every rank).

    `process_fn(local_frames) -> ctypes array of CdFrameResult` is the per-GPU hot path:
    perception_amd.capi.Context.process_batch[_device] in production; the CPU tests inject
    an oracle-backed function to exercise the sharding/gather logic under gloo."""

    def __init__(self, process_fn, dist=None, device=None):
        self.process_fn = process_fn
        self.dist = dist
        self.device = device

    def run(self, local_frames, n_frames_total):
        res = self.process_fn(local_frames)
        local = capi.results_to_array(res).copy()
        return gather_records(local, n_frames_total, self.dist, self.device)
