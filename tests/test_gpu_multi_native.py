"""The NATIVE frame-per-GPU driver (perception_amd/cpp/cuboid_multi_gpu.cpp: one thread + cd_context per GPU,
ncclCommInitAll, ONE ncclAllGather of the cd_frame_result records per batch - SURVEY 8(e)) against the Python driver
(perception_amd/batch.py) on the same 7-frame batch: byte-identical gathered records.  Frames are independent in the
reference (ground_plane_segmentation.cpp:146,153: a queue-1 subscriber on one spinner), so the slices need no exchange.
This box has one GPU: RCCL runs with one rank (the collective is really issued), three ranks share device 0 with the
gather through host memory (RCCL refuses two ranks on one device); N = 2/4/8 over xGMI is the round-end driver's run."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DRIVER = os.path.join(ROOT, "perception_amd", "cpp", "cuboid_multi_gpu")
NF = 7   # odd: ragged shards


@pytest.fixture(scope="module")
def batch_files(tmp_path_factory):
    from perception_amd import pcd, synth, templates
    d = tmp_path_factory.mktemp("native_multi")
    frames = np.stack([synth.frame(i) for i in range(NF)], 0)
    frames.tofile(d / "frames.bin")
    pcd.write_pcd_ascii(str(d / "template.pcd"), templates.make_cuboid_template(**templates.DEFAULT_TEMPLATE))
    return d, frames


@pytest.fixture(scope="module")
def python_records(batch_files):
    """the Python driver's gathered records: perception_amd.batch.ShardedBatchRunner (one rank: the whole batch)"""
    from perception_amd import batch, capi, templates
    _, frames = batch_files
    ctx = capi.Context(max_points=frames.shape[1], max_frames=NF)
    try:
        ctx.set_template(0, templates.template_xyz32(**templates.DEFAULT_TEMPLATE))
        prm = capi.default_params()
        prm.rgb_offset = 12
        runner = batch.ShardedBatchRunner(lambda fr: ctx.process_batch(fr, prm)[0])
        return runner.run(frames, NF).tobytes()
    finally:
        ctx.close()


def _run(d, frames, extra):
    assert os.path.exists(DRIVER), "build it: make -C perception_amd/cpp"
    out = d / ("records_%s.bin" % "_".join(a.strip("-").replace(",", "") for a in extra))
    r = subprocess.run([DRIVER, "--frames", str(d / "frames.bin"), "--points", str(frames.shape[1]), "--template", str(d / "template.pcd"),
                        "--out", str(out)] + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    return line, open(out, "rb").read()


def test_one_rank_through_rccl_allgather(batch_files, python_records):
    from perception_amd import capi
    d, frames = batch_files
    line, rec = _run(d, frames, ["--gpus", "1", "--steps", "2", "--warmup", "1"])
    assert line["gather"] == "rccl" and line["n_gpus"] == 1 and line["frames"] == NF and line["ranks_identical"] is True
    assert line["record_bytes"] == capi.FRAME_RESULT_BYTES
    assert rec == python_records


def test_three_ranks_on_one_device_ragged_shards(batch_files, python_records):
    """3 ranks x (3, 2, 2) frames, three contexts and three host threads on device 0; every rank's gathered copy equals rank 0's
    (the driver exits with 6 otherwise) and the Python driver's"""
    d, frames = batch_files
    line, rec = _run(d, frames, ["--gpus", "3", "--devices", "0,0,0", "--gather", "host"])
    assert line["gather"] == "host" and line["n_gpus"] == 3 and line["ranks_identical"] is True
    assert rec == python_records


def test_more_ranks_than_frames_and_bad_device_lists(batch_files, python_records):
    d, frames = batch_files
    line, rec = _run(d, frames, ["--gpus", "9", "--devices", "0,0,0,0,0,0,0,0,0", "--gather", "host"])   # two ranks own no frame
    assert line["ranks_identical"] is True and rec == python_records
    r = subprocess.run([DRIVER, "--frames", str(d / "frames.bin"), "--points", str(frames.shape[1]), "--template", str(d / "template.pcd"),
                        "--gpus", "2", "--devices", "0,0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "one device per rank" in r.stderr
