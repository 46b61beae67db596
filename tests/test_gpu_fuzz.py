"""Seeded parameter sweep of the whole chain: random crop limits, leaf sizes, plane thresholds, cluster tolerances and size
bounds on the synthetic frames - every record, plane index list and label array identical to the oracle's.  The fixed
launch parameters never reach most of the branches the kernels have (single-pass crop bit fields of other widths, 2- and
4-pass sorts, cell tables of other densities, clusters at the size bounds)."""
import numpy as np
import pytest

from perception_amd import capi, synth

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-4


def _same_record(rg, ro):
    for k in ("status", "n_cropped", "n_voxels", "n_plane", "n_objects", "n_clusters", "ransac_iterations"):
        assert getattr(rg, k) == getattr(ro, k), k
    assert bytes(rg.plane) == bytes(ro.plane)
    for k in range(min(rg.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
        a, b = rg.clusters[k], ro.clusters[k]
        assert (a.size, a.iterations, a.converged, a.accepted) == (b.size, b.iterations, b.converged, b.accepted), k
        assert list(a.T) == list(b.T) and a.fitness == b.fitness, k
        assert np.linalg.norm(np.array(a.pose) - np.array(b.pose)) < POSE_TOL


def test_random_parameters_match_oracle(O, template):
    rng = np.random.RandomState(2026)
    ctx = capi.Context(max_points=synth.WIDTH * synth.HEIGHT, max_frames=2)
    ctx.set_template(0, template)
    try:
        for trial in range(36):
            prm = capi.default_params()
            prm.rgb_offset = 12
            prm.leaf_size = float(rng.choice([0.0015, 0.003, 0.004, 0.005, 0.0075, 0.01, 0.02]))
            prm.crop_x_min, prm.crop_x_max = sorted(rng.uniform(-0.35, 0.35, 2).tolist())
            if prm.crop_x_max - prm.crop_x_min < 0.15:
                prm.crop_x_min, prm.crop_x_max = -0.2, 0.25
            prm.crop_z_min = float(rng.choice([0.0, 0.1, 0.3]))
            prm.crop_z_max = float(rng.uniform(0.6, 1.2))
            prm.plane_distance_threshold = float(rng.choice([0.005, 0.01, 0.015, 0.03]))
            prm.cluster_tolerance = float(rng.choice([0.008, 0.012, 0.02, 0.035, 0.05]))
            prm.cluster_min_size = int(rng.choice([1, 20, 100, 200]))
            prm.cluster_max_size = int(rng.choice([300, 1500, 25000]))
            if trial % 5 == 4:
                prm.icp_max_iterations = int(rng.choice([1, 3, 12]))
            frame = synth.frame(int(rng.randint(0, 64)))
            rg, pi, lb = ctx.process_frame(frame, prm, want_indices=True)
            o = O.process_frame(frame, prm, template, want_clouds=True)
            ro = o["result"]
            what = "trial %d leaf %g x [%g,%g] z [%g,%g] thr %g tol %g sizes [%d,%d]" % (
                trial, prm.leaf_size, prm.crop_x_min, prm.crop_x_max, prm.crop_z_min, prm.crop_z_max, prm.plane_distance_threshold,
                prm.cluster_tolerance, prm.cluster_min_size, prm.cluster_max_size)
            try:
                _same_record(rg, ro)
                assert np.array_equal(pi[:rg.n_plane], o["plane_inliers"])
                assert np.array_equal(lb[:rg.n_objects], o["labels"])
            except AssertionError as e:
                raise AssertionError(what + ": " + str(e))
    finally:
        ctx.close()


def test_random_parameters_batch_matches_oracle(O, template, monkeypatch):
    """The same on batches of 6 frames (the batch kernels: frame-interleaved chained scans, LDS clustering per frame, the
    pipelined ICP kernel forced so that its slots refill)."""
    rng = np.random.RandomState(77)
    monkeypatch.setenv("CUBOID_ICP_MODE", "pipe")
    monkeypatch.setenv("CUBOID_ICP_MAX_WG", "3")
    ctx = capi.Context(max_points=synth.WIDTH * synth.HEIGHT, max_frames=6)
    ctx.set_template(0, template)
    try:
        for trial in range(5):
            prm = capi.default_params()
            prm.rgb_offset = 12
            prm.leaf_size = float(rng.choice([0.004, 0.005, 0.0075, 0.01]))
            prm.crop_x_min, prm.crop_x_max = -float(rng.uniform(0.12, 0.3)), float(rng.uniform(0.12, 0.3))
            prm.crop_z_max = float(rng.uniform(0.7, 1.1))
            prm.cluster_tolerance = float(rng.choice([0.012, 0.02, 0.03]))
            prm.cluster_min_size = int(rng.choice([50, 200]))
            batch = np.stack([synth.frame(int(i)) for i in rng.randint(0, 64, 6)], 0)
            res, pi, lb = ctx.process_batch(batch, prm, want_indices=True)
            for f in range(6):
                o = O.process_frame(batch[f], prm, template, want_clouds=True)
                try:
                    _same_record(res[f], o["result"])
                    assert np.array_equal(pi[f][:res[f].n_plane], o["plane_inliers"])
                    assert np.array_equal(lb[f][:res[f].n_objects], o["labels"])
                except AssertionError as e:
                    raise AssertionError("batch trial %d frame %d leaf %g tol %g: %s" % (trial, f, prm.leaf_size, prm.cluster_tolerance, e))
    finally:
        ctx.close()
