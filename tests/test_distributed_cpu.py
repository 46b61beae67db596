"""Frame-per-rank sharding + record gather (perception_amd/batch.py) with 2 ranks over gloo on
the CPU.  The per-rank hot path is replaced by an oracle-backed stand-in (tests may call the
oracle); on GPUs the same driver runs with capi.Context.process_batch_device and the nccl
backend (= RCCL) - see bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from perception_amd import batch, capi

H, W, NF = 60, 80, 5     # small frames so the oracle finishes in seconds; odd count -> ragged shards


def _frames():
    from perception_amd import synth
    return np.stack([synth.frame(i, width=W, height=H) for i in range(NF)], 0)


def _oracle_fn(prm, tpl):
    from oracle import oracle_py as O

    def fn(local_frames):
        res = (capi.CdFrameResult * len(local_frames))()
        for i, f in enumerate(local_frames):
            res[i] = O.process_frame(f, prm, tpl)["result"]
        return res
    return fn


def _params():
    prm = capi.default_params()
    prm.leaf_size = 0.01
    prm.cluster_min_size = 20
    return prm


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from perception_amd import templates
    tpl = templates.template_xyz32(0.2, 0.1, 0.03, 0.01)
    frames = _frames()
    lo, hi = batch.shard_range(NF, rank, world)
    runner = batch.ShardedBatchRunner(_oracle_fn(_params(), tpl), dist=dist)
    rec = runner.run(frames[lo:hi], NF)
    q.put((rank, rec.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_batch():
    for n in (1, 5, 256, 2048, 7):
        for world in (1, 2, 3, 8):
            spans = [batch.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_two_rank_gather_equals_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from perception_amd import templates
    tpl = templates.template_xyz32(0.2, 0.1, 0.03, 0.01)
    ref = capi.results_to_array(_oracle_fn(_params(), tpl)(_frames())).tobytes()
    assert got[0] == got[1] == ref          # every rank holds the whole batch, in frame order
    recs = capi.results_from_array(np.frombuffer(ref, np.uint8).reshape(NF, capi.FRAME_RESULT_BYTES))
    assert all(r.n_voxels > 0 for r in recs)


def test_native_driver_argument_checks_without_a_gpu():
    """perception_amd/cpp/cuboid_multi_gpu (the native frame-per-GPU driver: one thread + context per GPU, ncclAllGather of the
    records) refuses, before it touches HIP or RCCL, a device list that puts two ranks on one device under RCCL, a list of the
    wrong length and a missing input; its slicing rule is batch.shard_range's (checked on the GPU box against the Python driver:
    tests/test_gpu_multi_native.py)."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "perception_amd", "cpp", "cuboid_multi_gpu")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "perception_amd", "cpp"), "cuboid_multi_gpu"], check=True, stdout=subprocess.DEVNULL)
    base = [exe, "--frames", "/nonexistent.bin", "--points", "10", "--template", "/nonexistent.pcd"]
    r = subprocess.run(base + ["--gpus", "2", "--devices", "0,0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "one device per rank" in r.stderr
    r = subprocess.run(base + ["--gpus", "3", "--devices", "0,1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "names 2 devices for 3 ranks" in r.stderr
    r = subprocess.run(base + ["--gpus", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2          # no such frames file
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage:" in r.stderr
