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
