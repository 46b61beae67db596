"""The N-GPU path of bench.py on real devices: frame-per-GPU sharding + ONE RCCL all_gather of the pose records per batch
(frames are independent: ground_plane_segmentation.cpp:146,153).  bench.py runs as a child process (fresh
HIP state per rank); the 2-rank case needs a 2-GPU box and is skipped on the 1-GPU box."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_one_rank_through_rccl_gather_is_verified():
    """World size 1 but the record gather goes through torch.distributed/nccl (= RCCL): same code path as N > 1."""
    out = _bench(["--gpus", "1", "--frames", "24", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], {"CUBOID_BENCH_FORCE_DIST": "1"})
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1
    assert out["verified"] is True
    assert out["icp"]["frames"] == 24 and out["single_frame_ms"]["median"] > 0


def test_two_ranks_nccl_gather():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs on the box (the round-end scaling run covers N = 2/4/8)")
    out = _bench(["--gpus", "2", "--frames", "24", "--steps", "4", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2
    assert out["verified"] is True
    assert out["icp"]["frames"] == 48          # every rank's records arrived, in frame order
    assert "cpu_baseline" not in out           # N = 1 only


def _rank_on_gpu0(rank, world, port, q):
    """one rank of the sharded path: HIP compute on GPU 0, its contiguous slice of the batch, record gather over gloo"""
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401  (before libcuboid_hip.so: see conftest)
    import torch.distributed as dist
    from perception_amd import batch, capi, synth, templates
    dist.init_process_group("gloo", rank=rank, world_size=world)
    NF = 7                                                      # odd: ragged shards
    frames = np.stack([synth.frame(i) for i in range(NF)], 0)
    lo, hi = batch.shard_range(NF, rank, world)
    ctx = capi.Context(max_points=frames.shape[1], max_frames=max(hi - lo, 1))
    ctx.set_template(0, templates.template_xyz32(**templates.DEFAULT_TEMPLATE))
    prm = capi.default_params()
    prm.rgb_offset = 12
    runner = batch.ShardedBatchRunner(lambda fr: ctx.process_batch(fr, prm)[0], dist=dist)
    rec = runner.run(frames[lo:hi], NF)
    q.put((rank, rec.tobytes()))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_two_ranks_share_one_gpu_gather_over_gloo():
    """The N > 1 data path on the hardware this box has: two rank processes, each running the HIP chain on its contiguous
    slice of a 7-frame batch (both on GPU 0 - RCCL refuses two ranks on one device, so the record gather goes over gloo;
    the 8-GPU RCCL run is the driver's) - every rank ends up with the whole batch in frame order, byte-identical to one
    process doing all 7 frames."""
    import socket
    import numpy as np
    import torch.multiprocessing as mp
    from perception_amd import capi, synth, templates
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_on_gpu0, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    frames = np.stack([synth.frame(i) for i in range(7)], 0)
    ctx = capi.Context(max_points=frames.shape[1], max_frames=7)
    try:
        ctx.set_template(0, templates.template_xyz32(**templates.DEFAULT_TEMPLATE))
        prm = capi.default_params()
        prm.rgb_offset = 12
        ref = capi.results_to_array(ctx.process_batch(frames, prm)[0]).tobytes()
    finally:
        ctx.close()
    assert got[0] == got[1] == ref
