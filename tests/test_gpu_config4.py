"""BASELINE config 4 AT ITS SIZE on the hardware this box has: 2048 frames, frame-per-GPU sharding, one gather of the pose records.

The reference processes frames one at a time and independently (ground_plane_segmentation.cpp:146,153: a queue-1 subscriber on
one spinner; RANSAC is re-seeded per frame), so the batch is cut into contiguous slices with no data-path collective
(DESIGN section 7).  No session has a multi-GPU node, so the eight ranks are rehearsed on ONE device:
  (a) the native driver, perception_amd/cpp/cuboid_multi_gpu: 8 rank threads x 256 frames, eight contexts on device 0, the
      gather through host memory (RCCL refuses two ranks on one device);
  (b) the Python driver, perception_amd.batch.ShardedBatchRunner: 4 rank processes x 512 frames (the box allows six GPU
      processes at most), each as two 256-frame calls, the record gather over gloo.
Checked: the gathered records are in frame order (every record carries its own frame's counts), every rank holds the same
2048 records, the two drivers agree byte for byte, and every 16th frame equals the CPU oracle.  What over xGMI / RCCL with
N > 1 remains untested is the collective itself.  Numbers (memory, time) go to gpurun_out/config4_rehearsal.json."""
import json
import os
import socket
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

DRIVER = os.path.join(ROOT, "perception_amd", "cpp", "cuboid_multi_gpu")
NF, PER_GPU, NPTS = 2048, 256, 640 * 480


def _py_rank(rank, world, port, path, q):
    """one rank process of the Python driver: its contiguous slice of the mapped batch, HIP on GPU 0, gather over gloo"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401  (before libcuboid_hip.so: see conftest)
    import torch.distributed as dist
    from perception_amd import batch, capi, templates
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = np.memmap(path, dtype=np.float32, mode="r", shape=(NF, NPTS, 4))
    lo, hi = batch.shard_range(NF, rank, world)
    ctx = capi.Context(max_points=NPTS, max_frames=PER_GPU)
    ctx.set_template(0, templates.template_xyz32(**templates.DEFAULT_TEMPLATE))
    prm = capi.default_params()
    prm.rgb_offset = 12

    def process(fr):   # a rank's slice as 256-frame calls: what one GPU gets per step in config 4
        out = (capi.CdFrameResult * len(fr))()
        arr = capi.results_to_array(out)
        for a in range(0, len(fr), PER_GPU):
            res, _, _ = ctx.process_batch(np.ascontiguousarray(fr[a:a + PER_GPU]), prm)
            arr[a:a + PER_GPU] = capi.results_to_array(res)
        return out

    t0 = time.perf_counter()
    rec = batch.ShardedBatchRunner(process, dist=dist).run(frames[lo:hi], NF)
    dt = time.perf_counter() - t0
    free, total = torch.cuda.mem_get_info(0)
    q.put((rank, rec.tobytes(), dt, (total - free) / 1e9))
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_2048_frames_as_eight_slices_native_and_python_drivers_vs_oracle(O, template, tmp_path):
    from perception_amd import batch, capi, pcd, synth, templates
    path = "/dev/shm/cuboid_config4_%d.bin" % os.getpid()     # memory-backed: mapped by the native driver and by every rank process
    t_render = time.perf_counter()
    try:
        frames = np.memmap(path, dtype=np.float32, mode="w+", shape=(NF, NPTS, 4))

        def work(i):
            frames[i] = synth.frame(i)

        with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:     # rendered on threads, before any HIP call of this test
            list(ex.map(work, range(NF)))
        frames.flush()
        t_render = time.perf_counter() - t_render
        tpl_path = str(tmp_path / "template.pcd")
        pcd.write_pcd_ascii(tpl_path, templates.make_cuboid_template(**templates.DEFAULT_TEMPLATE))

        # (a) native: eight rank threads x 256 frames on device 0
        assert os.path.exists(DRIVER), "build it: make -C perception_amd/cpp"
        out = str(tmp_path / "records_native.bin")
        t0 = time.perf_counter()
        r = subprocess.run([DRIVER, "--frames", path, "--points", str(NPTS), "--template", tpl_path, "--out", out, "--gpus", "8",
                            "--devices", "0,0,0,0,0,0,0,0", "--gather", "host", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=900)
        t_native = time.perf_counter() - t0
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 8 and line["frames"] == NF and line["ranks_identical"] is True and line["record_bytes"] == capi.FRAME_RESULT_BYTES
        native = np.fromfile(out, dtype=np.uint8).reshape(NF, capi.FRAME_RESULT_BYTES)

        # (b) Python: four rank processes x 512 frames, gloo gather
        import torch.multiprocessing as mp
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        world = 4
        mpc = mp.get_context("spawn")
        q = mpc.Queue()
        t0 = time.perf_counter()
        procs = [mpc.Process(target=_py_rank, args=(rk, world, port, path, q)) for rk in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=900) for _ in range(world)]
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        t_python = time.perf_counter() - t0
        by_rank = {g[0]: g for g in got}
        for rk in range(1, world):
            assert by_rank[rk][1] == by_rank[0][1], "rank %d holds other records than rank 0" % rk
        python = np.frombuffer(by_rank[0][1], dtype=np.uint8).reshape(NF, capi.FRAME_RESULT_BYTES)
        assert np.array_equal(native, python), "the native and the Python driver disagree"

        # frame order + the oracle on every 16th frame (frame-parallel on host threads)
        recs = capi.results_from_array(native)
        sample = list(range(0, NF, 16))
        prm = capi.default_params()
        prm.rgb_offset = 12
        O.lib()
        with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
            want = list(ex.map(lambda f: O.process_frame(np.asarray(frames[f]), prm, template, nn_mode=1)["result"], sample))
        from test_gpu_timed_path import assert_record_matches_oracle
        for f, ro in zip(sample, want):
            assert_record_matches_oracle(recs[f], ro, ("frame", f))
        # (a slice boundary that was off by one would shift every record after it: the counts of neighbouring frames differ)
        assert len({(recs[f].n_cropped, recs[f].n_voxels) for f in range(NF)}) > NF // 2
        ncl = sum(min(recs[f].n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME) for f in range(NF))

        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "config4_rehearsal.json"), "w") as fo:
            json.dump({"frames": NF, "clusters": ncl, "render_s": round(t_render, 1),
                       "native": {"ranks": 8, "frames_per_rank": PER_GPU, "devices": "0 x 8", "gather": "host", "wall_s": round(t_native, 1),
                                  "step_ms": line["ms_per_step"], "hbm_used_gb_all_ranks": line.get("hbm_used_gb_max")},
                       "python": {"ranks": world, "frames_per_rank": NF // world, "calls_per_rank": NF // world // PER_GPU, "gather": "gloo",
                                  "wall_s": round(t_python, 1), "rank_s": [round(by_rank[rk][2], 1) for rk in range(world)],
                                  "hbm_used_gb_all_ranks": round(max(by_rank[rk][3] for rk in range(world)), 2)},
                       "oracle_sample": len(sample), "byte_identical": True}, fo, indent=1)
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
