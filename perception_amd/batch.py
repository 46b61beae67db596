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


def gather_records(local_records, n_frames_total, dist=None, device=None, group=None):
    """All-gather per-frame records.  local_records: uint8 array (F_local, FRAME_RESULT_BYTES).
    Returns uint8 array (n_frames_total, FRAME_RESULT_BYTES) on every rank, frame order.
    device=None gathers host tensors (a gloo `group`); a device gathers through it (nccl = RCCL)."""
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
    dist.all_gather_into_tensor(out, t, group=group)
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


class BatchPipeline:
    """Several batches in flight on ONE GPU.

    The ICP kernel is a persistent launch whose workgroups finish at very different times (a cluster needs 6...100
    iterations), so its tail leaves CUs idle; the front end of the next batch is HBM-bound streaming work that fits
    there.  Each in-flight batch gets its own context (stream + device arena + pinned mirrors) and its own host
    thread - the C-ABI call is synchronous like the PCL calls it replaces, ctypes drops the GIL while it runs - and
    the GPU overlaps the streams.  Results are identical to processing the batches one after another.
    The streams must land on different hardware queues: export GPU_MAX_HW_QUEUES=8 (HIP's default of 4 is not enough:
    streams that share a queue run one after the other; 16 for multi-template batches, whose contexts have two more
    streams each) before the HIP runtime starts, as bench.py does.

    submit() returns a Future of (records uint8[F, FRAME_RESULT_BYTES], CdTiming)."""

    def __init__(self, max_points, max_frames, templates_by_slot, device_id=0, inflight=2):
        from concurrent.futures import ThreadPoolExecutor
        self.inflight = max(1, int(inflight))
        self.contexts = [capi.Context(max_points=max_points, max_frames=max_frames, device_id=device_id) for _ in range(self.inflight)]
        for cx in self.contexts:
            for slot, xyz in templates_by_slot.items():
                cx.set_template(slot, xyz)
        self._results = [(capi.CdFrameResult * max_frames)() for _ in range(self.inflight)]
        self._busy = [None] * self.inflight
        self._pool = ThreadPoolExecutor(self.inflight)

    def _run(self, i, d_ptr, stride, n_points, n_frames, prm, host=False):
        cx, res = self.contexts[i], self._results[i]
        if host:    # host-fed: the H2D copy is part of the call, on the context's own stream - it overlaps the other contexts' kernels
            cx.process_batch_host_ptr(d_ptr, stride, n_points, n_frames, prm, results=res)
        else:
            cx.process_batch_device(d_ptr, stride, n_points, n_frames, prm, results=res)
        return capi.results_to_array(res)[:n_frames].copy(), cx.timing()

    def submit(self, d_ptr, stride, n_points, n_frames, prm, host=False):
        """Queue one batch that is already resident in device memory (d_ptr: device pointer of F x N records), or - host=True -
        one that sits in HOST memory (pinned for full PCIe rate): what a ROS callback has (gps.cpp:43-49)."""
        # a context runs one batch at a time: take a free one, else wait for the first to finish (batches differ in length: a
        # fixed rotation would hold the submission behind the slowest)
        from concurrent.futures import FIRST_COMPLETED, wait
        free = [j for j in range(self.inflight) if self._busy[j] is None or self._busy[j].done()]
        if not free:
            wait(self._busy, return_when=FIRST_COMPLETED)
            free = [j for j in range(self.inflight) if self._busy[j].done()]
        i = free[0]
        fut = self._pool.submit(self._run, i, d_ptr, stride, n_points, n_frames, prm, host)
        self._busy[i] = fut
        return fut

    def close(self):
        self._pool.shutdown(wait=True)
        for cx in self.contexts:
            cx.close()
        self.contexts = []
