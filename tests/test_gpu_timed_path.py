"""Parity of the path bench.py TIMES, at the size it times it.

BASELINE config 3 = 256 synthetic frames per GPU through perception_amd.batch.BatchPipeline with three batches in
flight; the ICP stage auto-selects k_icp_pipe, whose grid is min(clusters, CUs) persistent workgroups: with ~520
clusters on 256 workgroups every workgroup refills its two slots from the queue (pipe_refill into a used slot, the
PH_FIT -> refill hand-over, two live slots per workgroup).  Every record of every batch is compared with the CPU
oracle (reference chain: object_detection/src/object_pose_detection.cpp:270-413, one ICP per cluster :376-413).
The cheap variant caps the ICP grid at 2 workgroups (CUBOID_ICP_MAX_WG) so that the refill paths run in every CI run."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from perception_amd import capi, synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4   # BASELINE.json north_star


def _oracle_records(O, frames, prm, tpl, threads=16):
    """oracle records of `frames`, frame-parallel on host threads (ctypes releases the GIL)."""
    O.lib()
    with ThreadPoolExecutor(max(1, min(threads, os.cpu_count() or 1))) as ex:
        return list(ex.map(lambda f: O.process_frame(f, prm, tpl, nn_mode=1)["result"], frames))


def assert_record_matches_oracle(rg, ro, tag):
    assert (rg.status, rg.n_cropped, rg.n_voxels, rg.n_plane, rg.n_objects, rg.n_clusters, rg.ransac_iterations) == \
           (ro.status, ro.n_cropped, ro.n_voxels, ro.n_plane, ro.n_objects, ro.n_clusters, ro.ransac_iterations), tag
    assert np.array_equal(np.array(rg.plane, np.float32).view(np.uint32), np.array(ro.plane, np.float32).view(np.uint32)), tag
    for k in range(min(ro.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
        a, b = rg.clusters[k], ro.clusters[k]
        assert (a.size, a.iterations, a.converged, a.accepted) == (b.size, b.iterations, b.converged, b.accepted), (tag, k)
        assert list(a.T) == list(b.T), (tag, k, "final transformation not bit-identical")
        assert a.fitness == b.fitness, (tag, k)
        assert np.linalg.norm(np.array(a.pose) - np.array(b.pose)) < POSE_TOL, (tag, k)


def _render(lo, n):
    with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
        return np.stack(list(ex.map(synth.frame, range(lo, lo + n))), 0)


def test_bench_config3_five_batches_in_flight_vs_oracle(O, template):
    """Exactly what bench.py times: 256 frames per batch, default ICP mode (auto -> k_icp_pipe), BatchPipeline(inflight=5) -
    with four or more calls on the device a launch keeps FOUR clusters in flight per workgroup on 128 workgroups - eight
    consecutive batches (so contexts are reused while the others are busy); every record vs the oracle."""
    import torch
    from perception_amd import batch
    F = 256
    prm = capi.default_params()
    prm.rgb_offset = 12
    sets = [_render(0, F), _render(F, F)]
    want = [_oracle_records(O, s, prm, template) for s in sets]
    ncl = sum(min(r.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME) for r in want[0])
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    assert ncl >= 2 * n_cu - 8, "the batch must hold ~2 clusters per CU so that k_icp_pipe refills its slots (got %d on %d CUs)" % (ncl, n_cu)
    dev = [torch.from_numpy(s).cuda() for s in sets]
    torch.cuda.synchronize()
    N = sets[0].shape[1]
    pipe = batch.BatchPipeline(N, F, {0: template}, inflight=5)
    try:
        order = [0, 1, 0, 1, 1, 0, 0, 1]
        futs = [pipe.submit(dev[i].data_ptr(), 16, N, F, prm) for i in order]
        for step, (i, fut) in enumerate(zip(order, futs)):
            rec, tim = fut.result()
            assert tim.icp_kernel_launches == 1, "batch mode must run ONE persistent ICP launch (k_icp_pipe)"
            got = capi.results_from_array(rec)
            for f in range(F):
                assert_record_matches_oracle(got[f], want[i][f], ("step", step, "frame", i * F + f))
    finally:
        pipe.close()


@pytest.mark.parametrize("slots", ["", "1", "3", "4"])
@pytest.mark.parametrize("max_wg", ["1", "2", "5"])
def test_pipe_slot_refill_with_capped_grid(O, template, max_wg, slots, monkeypatch):
    """k_icp_pipe on 1, 2 or 5 workgroups with ~20 clusters: every slot is refilled several times, the slots of a
    workgroup hold clusters at different iterations, and the last clusters leave slots exhausted while another
    still works - with the slot count of the regime rule (two: the call has the GPU to itself) and forced to 1, 3 and 4
    (four is what a launch uses that shares the GPU with three or more calls).  Records must equal the oracle's, and the
    sliced driver's bytes."""
    if slots:
        monkeypatch.setenv("CUBOID_ICP_SLOTS", slots)
    idx = list(range(40, 50))
    frames = np.stack([synth.frame(i) for i in idx], 0)
    prm = capi.default_params()
    want = _oracle_records(O, frames, prm, template)
    assert sum(r.n_clusters for r in want) >= 12
    monkeypatch.setenv("CUBOID_ICP_MODE", "pipe")
    monkeypatch.setenv("CUBOID_ICP_MAX_WG", max_wg)
    ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
    try:
        ctx.set_template(0, template)
        for rep in range(2):                      # the second call reuses the queue / state buffers
            res, _, _ = ctx.process_batch(frames, prm)
            assert ctx.timing().icp_kernel_launches == 1
            for f in range(len(frames)):
                assert_record_matches_oracle(res[f], want[f], (max_wg, rep, idx[f]))
        pipe_bytes = capi.results_to_array(res).copy()
    finally:
        ctx.close()
    monkeypatch.setenv("CUBOID_ICP_MODE", "sliced")
    monkeypatch.delenv("CUBOID_ICP_MAX_WG")
    ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
    try:
        ctx.set_template(0, template)
        res, _, _ = ctx.process_batch(frames, prm)
        assert np.array_equal(capi.results_to_array(res), pipe_bytes)
    finally:
        ctx.close()


@pytest.mark.parametrize("lowprio", ["", "2"])
@pytest.mark.parametrize("slots", ["2", "4"])
def test_pipe_handover_of_running_clusters(O, template, slots, lowprio, monkeypatch):
    """A launch that has the GPU to itself lets workgroups that ran out of clusters take over RUNNING ones from workgroups
    that still have several (k_icp.hip, pipe_give / pipe_wait_for_cluster: the cluster's points and neighbour indices cross to
    another CU, possibly another XCD, in the middle of its ICP).  Deterministic, not a matter of timing: with
    CUBOID_ICP_DON_IDLE=1 workgroup 0 never takes from the queue, so every cluster it runs reached it by hand-over, and with
    31 clusters on 8 workgroups the other seven hold several clusters each for most of the launch - at least one hand-over
    per call is certain (cd_timing.icp_handovers).  The records must be the oracle's and byte-identical to a launch without
    hand-overs.  lowprio = 2: the launch goes to the context's side stream (the control block must be zeroed before that
    stream is released - ADVICE r4)."""
    idx = list(range(60, 76))
    frames = np.stack([synth.frame(i) for i in idx], 0)
    prm = capi.default_params()
    want = _oracle_records(O, frames, prm, template)
    assert sum(r.n_clusters for r in want) >= 24
    monkeypatch.setenv("CUBOID_ICP_MODE", "pipe")
    monkeypatch.setenv("CUBOID_ICP_MAX_WG", "8")
    monkeypatch.setenv("CUBOID_ICP_SLOTS", slots)
    if lowprio:
        monkeypatch.setenv("CUBOID_ICP_LOWPRIO", lowprio)
    got = {}
    for donate in ("1", "0"):
        monkeypatch.setenv("CUBOID_ICP_DONATE", donate)
        monkeypatch.setenv("CUBOID_ICP_DON_IDLE", "1" if donate == "1" else "0")
        ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
        try:
            ctx.set_template(0, template)
            per_call = []
            for rep in range(3):                  # (the control block and the mailbox are reused from call to call)
                res, _, _ = ctx.process_batch(frames, prm)
                t = ctx.timing()
                assert t.icp_kernel_launches == 1 and t.icp_handover_lost == 0
                per_call.append(t.icp_handovers)
                for f in range(len(frames)):
                    assert_record_matches_oracle(res[f], want[f], (donate, rep, idx[f]))
            got[donate] = (capi.results_to_array(res).copy(), per_call)
        finally:
            ctx.close()
    assert got["0"][1] == [0, 0, 0]
    assert all(h >= 1 for h in got["1"][1]), "the idle workgroup can only have worked on hand-overs: %s" % got["1"][1]
    assert np.array_equal(got["0"][0], got["1"][0])


def test_pipe_handover_that_loses_a_cluster_is_reported(template, monkeypatch):
    """Fault injection (CUBOID_ICP_DON_FAULT=1): the first donor claims its mailbox entry and never publishes it, so the cluster
    has left its workgroup and reaches nobody.  The taker gives up on the entry, raises the error word, every waiter leaves,
    and the HOST fails the call with CD_ERR_DEVICE and cd_timing.icp_handover_lost = 1 - not a silent record with done = 0.
    The context works again afterwards."""
    idx = list(range(60, 76))
    frames = np.stack([synth.frame(i) for i in idx], 0)
    prm = capi.default_params()
    monkeypatch.setenv("CUBOID_ICP_MODE", "pipe")
    monkeypatch.setenv("CUBOID_ICP_MAX_WG", "8")
    monkeypatch.setenv("CUBOID_ICP_DONATE", "1")
    monkeypatch.setenv("CUBOID_ICP_DON_IDLE", "1")
    monkeypatch.setenv("CUBOID_ICP_DON_FAULT", "1")
    ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
    monkeypatch.setenv("CUBOID_ICP_DON_FAULT", "0")
    good = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
    try:
        ctx.set_template(0, template)
        good.set_template(0, template)
        with pytest.raises(capi.CuboidError) as e:
            ctx.process_batch(frames, prm)
        assert e.value.status == capi.CD_ERR_DEVICE and "hand-over lost a cluster" in str(e.value)
        assert ctx.timing().icp_handover_lost == 1
        res, _, _ = good.process_batch(frames, prm)          # the device is fine: a context without the fault gives the usual records
        assert good.timing().icp_handover_lost == 0 and good.timing().icp_handovers >= 1
    finally:
        ctx.close()
        good.close()


def test_pipe_big_handover_of_running_clusters(O, monkeypatch):
    """The same hand-over in k_icp_pipe_big (the template stays in global memory: the reference's 21 400-point six-face cuboid):
    records with hand-overs forced on equal the oracle's and the bytes of a launch without them."""
    from conftest import GOLDEN
    from perception_amd import pcd
    big = pcd.read_xyz(os.path.join(GOLDEN, "template_cuboid_L200_W100_H75.pcd")).astype(np.float32)
    idx = list(range(60, 72))
    frames = np.stack([synth.frame(i) for i in idx], 0)
    prm = capi.default_params()
    want = _oracle_records(O, frames, prm, big)
    monkeypatch.setenv("CUBOID_ICP_MODE", "pipe")
    monkeypatch.setenv("CUBOID_ICP_MAX_WG", "8")
    got = {}
    for donate in ("1", "0"):
        monkeypatch.setenv("CUBOID_ICP_DONATE", donate)
        monkeypatch.setenv("CUBOID_ICP_DON_IDLE", "1" if donate == "1" else "0")   # workgroup 0 only ever works on hand-overs: not a matter of timing
        ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
        try:
            ctx.set_template(0, big)
            handovers = 0
            for rep in range(2):
                res, _, _ = ctx.process_batch(frames, prm)
                handovers += ctx.timing().icp_handovers
                for f in range(len(frames)):
                    assert_record_matches_oracle(res[f], want[f], ("big", donate, rep, idx[f]))
            got[donate] = (capi.results_to_array(res).copy(), handovers)
        finally:
            ctx.close()
    assert got["0"][1] == 0 and got["1"][1] > 0, "no cluster changed workgroup"
    assert np.array_equal(got["0"][0], got["1"][0])


@pytest.mark.parametrize("switch", ["CUBOID_COPY_KERNELS", "CUBOID_ZERO_ONCE", "CUBOID_CROP_DIRECT", "CUBOID_CENTROID_LANES", "CUBOID_MIRROR_READS",
                                    "CUBOID_MIRROR_WRITES", "CUBOID_ICP_DIRECT"])
def test_plumbing_switches_leave_the_records_unchanged(template, switch, monkeypatch):
    """Round 5 moved the small pinned <-> device transfers from hipMemcpyAsync to copy kernels on the context's stream (batched:
    one launch per stage), all zero fills of a fused call into one launch, the cropped points out of the arena (run records point
    into the input), the centroid sums onto one lane per voxel, and let kernels read the host's pinned per-frame words and write
    the FrameState mirror and the ICP results themselves.  Each has a switch back to the previous form: the records of a
    fused batch call must be the same bytes either way (the default form is compared with the oracle by every other test)."""
    F = 6
    frames = _render(40, F)
    prm = capi.default_params()
    prm.rgb_offset = 12

    def run():
        ctx = capi.Context(max_points=frames.shape[1], max_frames=F)
        try:
            ctx.set_template(0, template)
            res, plane_idx, labels = ctx.process_batch(frames, prm, want_indices=True)
            return (capi.results_to_array(res).copy(), [plane_idx[f, :res[f].n_plane].copy() for f in range(F)],
                    [labels[f, :res[f].n_objects].copy() for f in range(F)])
        finally:
            ctx.close()

    a = run()
    monkeypatch.setenv(switch, "0")     # (read when a context is created)
    b = run()
    assert np.array_equal(a[0], b[0])
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert np.array_equal(x, y)
