"""GPU tests at BASELINE.json's larger sizes, through size-independent properties plus one
oracle comparison at 1 M points (config 5's frame size)."""
import os

import numpy as np
import pytest

from perception_amd import capi, synth, templates

pytestmark = pytest.mark.gpu


def test_batch_of_32_equals_32_single_frames(template):
    """A frame's record must not depend on the batch it travels in (frames are independent:
    that is what makes frame-per-GPU sharding exact)."""
    F = 32
    frames = np.stack([synth.frame(100 + i) for i in range(F)], 0)
    prm = capi.default_params()
    prm.rgb_offset = 12
    ctx = capi.Context(max_points=frames.shape[1], max_frames=F)
    ctx.set_template(0, template)
    res, pi, lb = ctx.process_batch(frames, prm, want_indices=True)
    A = capi.results_to_array(res).copy()
    one = capi.Context(max_points=frames.shape[1], max_frames=1)
    one.set_template(0, template)
    for f in range(F):
        r1, p1, l1 = one.process_batch(frames[f:f + 1], prm, want_indices=True)
        assert np.array_equal(capi.results_to_array(r1)[0], A[f]), f
        assert np.array_equal(p1[0], pi[f]) and np.array_equal(l1[0], lb[f])
    # every frame found its plane and at least one cuboid, and ICP accepted most of them
    assert all(r.status == 0 and r.n_clusters >= 1 for r in res)
    acc = sum(r.clusters[k].accepted for r in res for k in range(r.n_clusters))
    assert acc >= 0.7 * sum(r.n_clusters for r in res)
    ctx.close()
    one.close()


def test_one_million_point_frame_matches_oracle(O, template):
    """1000x1000 virtual sensor (1 M points, the frame size of BASELINE config 5), 3 cuboids, two
    template slots; the second ICP target is the 75 mm-high cuboid template."""
    sc = synth.scene_for(7, k_obj=3)
    big = synth.render(sc, width=1000, height=1000)
    assert big.shape == (1000000, 4)
    prm = capi.default_params()
    prm.rgb_offset = 12
    ctx = capi.Context(max_points=big.shape[0], max_frames=1)
    ctx.set_template(0, template)
    tall = templates.template_xyz32(0.2, 0.1, 0.075, 0.002)
    ctx.set_template(1, tall)
    for slot, tpl in ((0, template), (1, tall)):
        prm.template_slot = slot
        res, pi, lb = ctx.process_batch(big[None], prm, want_indices=True)
        o = O.process_frame(big, prm, tpl, want_clouds=True)
        rg, ro = res[0], o["result"]
        assert (rg.n_cropped, rg.n_voxels, rg.n_plane, rg.n_objects, rg.n_clusters) == \
               (ro.n_cropped, ro.n_voxels, ro.n_plane, ro.n_objects, ro.n_clusters)
        assert np.array_equal(pi[0][:rg.n_plane], o["plane_inliers"])
        assert np.array_equal(lb[0][:rg.n_objects], o["labels"])
        for k in range(min(rg.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
            a, b = rg.clusters[k], ro.clusters[k]
            assert (a.size, a.iterations, a.converged) == (b.size, b.iterations, b.converged)
            assert list(a.T) == list(b.T) and a.fitness == b.fitness
    # template_slot = -1: every cluster against every loaded template, best (lowest) fitness wins
    prm.template_slot = -1
    res, _, _ = ctx.process_batch(big[None], prm)
    per = {}
    for slot, tpl in ((0, template), (1, tall)):
        prm.template_slot = slot
        per[slot] = O.process_frame(big, prm, tpl)["result"]
    rg = res[0]
    assert rg.n_clusters == per[0].n_clusters
    for k in range(min(rg.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
        want = min((0, 1), key=lambda s_: (per[s_].clusters[k].fitness, s_))
        a, b = rg.clusters[k], per[want].clusters[k]
        assert a.template_slot == want
        assert (a.size, a.iterations, a.converged) == (b.size, b.iterations, b.converged)
        assert list(a.T) == list(b.T) and a.fitness == b.fitness
    ctx.close()


def test_capacity_and_argument_errors(template):
    ctx = capi.Context(max_points=1000, max_frames=2)
    prm = capi.default_params()
    with pytest.raises(capi.CuboidError) as e:
        ctx.process_batch(np.zeros((1, 2000, 4), np.float32), prm)
    assert e.value.status == capi.CD_ERR_CAPACITY
    with pytest.raises(capi.CuboidError) as e:
        ctx.process_batch(np.zeros((3, 100, 4), np.float32), prm)
    assert e.value.status == capi.CD_ERR_CAPACITY
    with pytest.raises(capi.CuboidError) as e:          # no template in the slot
        ctx.icp(3, template[:100], prm)
    assert e.value.status == capi.CD_ERR_NO_TEMPLATE
    prm.leaf_size = 0.0
    with pytest.raises(capi.CuboidError) as e:
        ctx.process_batch(np.zeros((1, 100, 4), np.float32), prm)
    assert e.value.status == capi.CD_ERR_INVALID_ARG
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sliced", "cluster", "pipe"])
def test_every_icp_driver_gives_the_same_bits(O, template, mode, monkeypatch):
    """The three ICP drivers (sliced multi-launch, one cluster per workgroup, two-slot pipeline) and both
    template layouts must give bit-identical results; the mode is read when the context is created."""
    monkeypatch.setenv("CUBOID_ICP_MODE", mode)
    frames = np.stack([synth.frame(i) for i in (0, 5, 9)], 0)
    prm = capi.default_params()
    ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
    try:
        ctx.set_template(0, template)
        res, _, _ = ctx.process_batch(frames, prm)
        for f in range(len(frames)):
            ro = O.process_frame(frames[f], prm, template)["result"]
            assert res[f].n_clusters == ro.n_clusters
            for k in range(min(ro.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
                a, b = res[f].clusters[k], ro.clusters[k]
                assert (a.size, a.iterations, a.converged, a.accepted) == (b.size, b.iterations, b.converged, b.accepted), (mode, f, k)
                assert list(a.T) == list(b.T) and a.fitness == b.fitness, (mode, f, k)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["generated_10700", "reference_21400"])
@pytest.mark.parametrize("mode", ["sliced", "cluster", "pipe"])
def test_template_larger_than_lds_in_batch_mode(O, mode, which, monkeypatch):
    """A 10 700-point template - and the reference's own 21 400-point six-face template_cuboid_L200_W100_H75.pcd - do not
    fit the LDS image: 'pipe' runs them through k_icp_pipe_big (the pipelined kernel with the template left in global
    memory: grid walk over the cell-sorted copy, patch search over the k-d ordered one, two box levels), 'cluster' through
    the chunked search of k_icp_cluster, 'sliced' through the chunked multi-launch driver; results match the oracle bit for
    bit in all three."""
    from conftest import GOLDEN
    from perception_amd import pcd
    monkeypatch.setenv("CUBOID_ICP_MODE", mode)
    if which == "generated_10700":
        big = templates.template_xyz32(length=0.2, width=0.1, height=0.075, density=0.002)
    else:
        big = pcd.read_xyz(os.path.join(GOLDEN, "template_cuboid_L200_W100_H75.pcd")).astype(np.float32)
        assert len(big) == 21400
    assert len(big) > 7616
    frames = np.stack([synth.frame(i) for i in (0, 5)], 0)
    prm = capi.default_params()
    ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
    try:
        ctx.set_template(0, big)
        res, _, _ = ctx.process_batch(frames, prm)
        for f in range(len(frames)):
            ro = O.process_frame(frames[f], prm, big)["result"]
            assert res[f].n_clusters == ro.n_clusters
            for k in range(min(ro.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
                a, b = res[f].clusters[k], ro.clusters[k]
                assert a.iterations == b.iterations and list(a.T) == list(b.T) and a.fitness == b.fitness, (mode, f, k)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_batches_in_flight_give_the_same_records(template):
    """BatchPipeline: three batches submitted to two contexts overlap on the GPU; every record must be byte-identical
    to the one a single context produces for the same batch."""
    import torch
    from perception_amd import batch
    sets = [np.stack([synth.frame(i) for i in idx], 0) for idx in ((0, 1, 2), (3, 4, 5), (6, 7, 8))]
    prm = capi.default_params()
    N = sets[0].shape[1]
    ctx = capi.Context(max_points=N, max_frames=3)
    ctx.set_template(0, template)
    want = [capi.results_to_array(ctx.process_batch(s, prm)[0]).copy() for s in sets]
    ctx.close()
    dev = [torch.from_numpy(s).cuda() for s in sets]
    torch.cuda.synchronize()
    pipe = batch.BatchPipeline(N, 3, {0: template}, inflight=2)
    try:
        futs = [pipe.submit(d.data_ptr(), 16, N, 3, prm) for d in dev]
        for f, w in zip(futs, want):
            rec, _ = f.result()
            assert np.array_equal(rec, w)
    finally:
        pipe.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sliced", "cluster", "pipe"])
def test_irregular_templates_in_every_driver(O, mode, monkeypatch):
    """Real D435 clusters from the reference tree (irregular density, holes) as TEMPLATES: the uniform grid is sized
    from their median point spacing, cells hold anything from 0 to dozens of points, the seed balls reach outside the
    grid - the lane-per-query walk and the run-box search must still return the oracle's bits in every driver."""
    from conftest import GOLDEN, rot_xyz
    from perception_amd import pcd
    monkeypatch.setenv("CUBOID_ICP_MODE", mode)
    rng = np.random.default_rng(77)
    ctx = capi.Context(max_points=8192, max_frames=1)
    try:
        for slot, name in enumerate(("marker_ascii.pcd", "screwdriver_ascii.pcd", "eraser_ascii.pcd")):
            tpl = pcd.read_xyz(os.path.join(GOLDEN, name)).astype(np.float32)
            ctx.set_template(slot, tpl)
            for trial in range(4):
                keep = rng.random(len(tpl)) < rng.uniform(0.4, 1.0)
                R = rot_xyz(*(rng.uniform(-0.08, 0.08, 3)))
                c = tpl.mean(0)
                src = ((tpl[keep] - c) @ R.T + c + rng.uniform(-0.01, 0.01, 3) + rng.normal(0, 0.0005, (keep.sum(), 3))).astype(np.float32)
                if trial == 3:
                    src = src + np.float32(0.25)          # far outside the template's grid: every seed ball is huge at first
                prm = capi.default_params()
                prm.icp_max_iterations = 60
                st, res, _ = ctx.icp(slot, src, prm)
                s0, r0, _ = O.icp(tpl, src, prm, nn_mode=1)
                assert st == s0 == 0
                assert (res.iterations, res.converged) == (r0.iterations, r0.converged), (mode, name, trial)
                assert list(res.T) == list(r0.T) and res.fitness == r0.fitness, (mode, name, trial)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cluster", "pipe"])
def test_whole_cluster_drivers_on_odd_sizes(O, template, mode, monkeypatch):
    """Pass boundaries of the whole-cluster kernels: sources of 3, 63, 64, 65, 127, 129, 1025 and 20 000 points (passes are
    64 consecutive points pulled by waves; the last one is ragged) must give the oracle's bits."""
    from conftest import rot_xyz
    monkeypatch.setenv("CUBOID_ICP_MODE", mode)
    rng = np.random.default_rng(3)
    ctx = capi.Context(max_points=32768, max_frames=1)
    try:
        ctx.set_template(0, template)
        for n in (3, 63, 64, 65, 127, 129, 1025, 20000):
            idx = rng.integers(0, len(template), n)
            R = rot_xyz(*(rng.uniform(-0.05, 0.05, 3)))
            src = (template[idx] @ R.T + rng.uniform(-0.004, 0.004, 3) + rng.normal(0, 0.0007, (n, 3))).astype(np.float32)
            prm = capi.default_params()
            prm.icp_max_iterations = 40
            st, res, _ = ctx.icp(0, src, prm)
            s0, r0, _ = O.icp(template, src, prm, nn_mode=1)
            assert st == s0, (mode, n)
            assert (res.iterations, res.converged) == (r0.iterations, r0.converged), (mode, n)
            assert list(res.T) == list(r0.T) and res.fitness == r0.fitness, (mode, n)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_template_arena_is_reclaimed(O, template):
    """cd_set_template on the same slot with ever larger templates: the space of the replaced ones is reclaimed (the arena
    holds 2^18 points; 45 templates of ~7000+ points would not fit side by side), other slots survive the re-packing, and
    ICP against the re-packed templates still gives the oracle's bits."""
    from conftest import rot_xyz
    rng = np.random.default_rng(5)
    ctx = capi.Context(max_points=8192, max_frames=1)
    try:
        small = templates.template_xyz32(0.05, 0.05, 0.03, 0.002)
        ctx.set_template(3, small)
        for i in range(45):
            grown = np.concatenate([template, template[: 64 * (i + 1)] + np.float32(0.3)])
            ctx.set_template(0, grown)
        prm = capi.default_params()
        prm.icp_max_iterations = 30
        for slot, tpl in ((0, grown), (3, small)):
            idx = rng.integers(0, len(tpl), 900)
            src = (tpl[idx] @ rot_xyz(0.02, -0.03, 0.04).T + np.float32(0.002)).astype(np.float32)
            st, res, _ = ctx.icp(slot, src, prm)
            s0, r0, _ = O.icp(tpl, src, prm, nn_mode=1)
            assert st == s0 == 0 and res.iterations == r0.iterations
            assert list(res.T) == list(r0.T) and res.fitness == r0.fitness
        with pytest.raises(capi.CuboidError) as e:          # genuinely too large
            ctx.set_template(1, np.zeros((300000, 3), np.float32))
        assert e.value.status == capi.CD_ERR_CAPACITY
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "pipe"])
def test_config5_one_million_points_five_templates(O, mode, monkeypatch):
    """BASELINE config 5 as SURVEY 8(d) writes it: a 1 M-point frame (1000 x 1000 virtual sensor), five cuboids of distinct
    dimensions, five templates - the three the reference ships dims for (one of them the committed 1700-point file with its
    origin at a corner), 150x150x50 (9375 points: does NOT fit the LDS image) and 100x100x100 at d = 0.002 - crops widened
    to the table, template_slot = -1: every cluster against every template, the lowest fitness wins.  mode 'pipe' forces
    what a batch of such frames takes by itself: ONE ICP stage of two persistent launches side by side - k_icp_pipe for the
    four template groups that fit LDS, k_icp_pipe_big for the 9375-point one."""
    from conftest import GOLDEN
    if mode == "pipe":
        monkeypatch.setenv("CUBOID_ICP_MODE", "pipe")
    from perception_amd import pcd
    frame = synth.frame_config5(0)
    assert frame.shape == (1000000, 4)
    tpls = []
    for k, (L, W, H, d) in enumerate(synth.CONFIG5_DIMS):
        if k == 2:
            tpls.append(pcd.read_xyz(os.path.join(GOLDEN, "template_cuboid_L200_W100_H75_3faces.pcd")).astype(np.float32))
        else:
            tpls.append(templates.template_xyz32(L, W, H, d))
    assert [len(t) for t in tpls] == [7250, 1700, 1700, 9375, 7500]
    prm = capi.default_params()
    prm.rgb_offset = 12
    prm.crop_x_min, prm.crop_x_max = -synth.CONFIG5_CROP_X, synth.CONFIG5_CROP_X
    prm.crop_z_max = prm.crop2_z_max = 1.2
    per = []
    for slot, t in enumerate(tpls):
        prm.template_slot = slot
        per.append(O.process_frame(frame, prm, t, all_clusters=16))
    assert per[0]["result"].n_clusters == 5
    ctx = capi.Context(max_points=frame.shape[0], max_frames=1)
    try:
        for slot, t in enumerate(tpls):
            ctx.set_template(slot, t)
        prm.template_slot = -1
        res, pi, lb = ctx.process_batch(frame[None], prm, want_indices=True)
        r, ro = res[0], per[0]["result"]
        assert (r.n_cropped, r.n_voxels, r.n_plane, r.n_objects, r.n_clusters) == (ro.n_cropped, ro.n_voxels, ro.n_plane, ro.n_objects, ro.n_clusters)
        assert np.array_equal(pi[0][:r.n_plane], per[0]["plane_inliers"]) and np.array_equal(lb[0][:r.n_objects], per[0]["labels"])
        picked = []
        for k in range(r.n_clusters):
            want = min(range(5), key=lambda s_: (per[s_]["clusters"][k].fitness, s_))
            a, b = r.clusters[k], per[want]["clusters"][k]
            assert a.template_slot == want, k
            assert (a.size, a.iterations, a.converged, a.accepted) == (b.size, b.iterations, b.converged, b.accepted), k
            assert list(a.T) == list(b.T) and a.fitness == b.fitness, k
            picked.append(want)
        assert len(set(picked)) >= 3      # the templates really compete: different cuboids pick different templates
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "pipe"])
@pytest.mark.parametrize("m", [7488, 7551, 7552, 7553, 7616, 7617, 65535, 65536])
def test_template_sizes_at_the_kernel_limits(O, m, mode, monkeypatch):
    """Template sizes either side of the kernels' limits: 7552 points (118 runs) is the last template whose image fits LDS
    beside four pipeline slots (k_icp_pipe; 7551 leaves one pad point in its last 64-point run, 7488 none), 7553 the first
    that stays in global memory (k_icp_pipe_big; 7616 / 7617 were that boundary with two slots), 65535 the last a 16-bit
    position can address and 65536 the first that takes the sliced driver.  Random subsets of a denser cuboid template, two
    frames; every size gives the oracle's bits."""
    monkeypatch.setenv("CUBOID_ICP_MODE", mode)
    dense = templates.template_xyz32(length=0.2, width=0.1, height=0.03, density=0.0006 if m > 8000 else 0.0018)
    assert len(dense) > m
    rng = np.random.default_rng(m)
    tpl = np.ascontiguousarray(dense[np.sort(rng.choice(len(dense), m, replace=False))])
    frames = np.stack([synth.frame(i) for i in (1, 4)], 0)
    prm = capi.default_params()
    ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
    try:
        ctx.set_template(0, tpl)
        res, _, _ = ctx.process_batch(frames, prm)
        for f in range(len(frames)):
            ro = O.process_frame(frames[f], prm, tpl)["result"]
            assert res[f].n_clusters == ro.n_clusters and ro.n_clusters > 0
            for k in range(min(ro.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
                a, b = res[f].clusters[k], ro.clusters[k]
                assert a.iterations == b.iterations and list(a.T) == list(b.T) and a.fitness == b.fitness, (m, mode, f, k)
    finally:
        ctx.close()


def _hard_geometry_cases(template):
    """(name, template, source) triples that leave the comfortable range of the search structures"""
    from conftest import rot_xyz
    rng = np.random.default_rng(11)
    R = rot_xyz(0.03, -0.02, 0.04)
    pick = lambda t, n: t[rng.integers(0, len(t), n)]
    cases = []
    # coordinates near 300 m: the moments leave the fast fixed-point conversion's range (|coordinate| >= 256), float spacing is 3e-5
    off = np.array([300.0, -280.0, 290.0], np.float32)
    cases.append(("300 m away", template + off, ((pick(template, 700) - template.mean(0)) @ R.T + template.mean(0) + 0.003).astype(np.float32) + off))
    # a flat template: the uniform grid is one cell thick
    g = np.arange(0, 0.12, 0.002, dtype=np.float32)
    flat = np.stack(np.meshgrid(g, g, [np.float32(0.5)], indexing="ij"), -1).reshape(-1, 3)
    cases.append(("flat template", flat, (pick(flat, 500) @ R.T + [0.002, -0.001, 0.004]).astype(np.float32)))
    # five template points: fewer than a run, fewer than the k-d search has lanes
    few = template[[0, 1000, 2000, 3000, 4000]]
    cases.append(("5-point template", few, (pick(template, 300) + np.float32(0.001)).astype(np.float32)))
    # every template point twice (exact ties on distance between different original indices), shuffled
    dup = np.concatenate([template[::2], template[::2]])[rng.permutation(2 * len(template[::2]))]
    cases.append(("duplicated points", dup, (pick(template, 900) @ R.T + np.float32(0.002)).astype(np.float32)))
    # source a metre away from the template: every query is far in every iteration until the centroids meet
    cases.append(("1 m apart", template, (pick(template, 600) @ R.T + [1.0, 0.5, -0.8]).astype(np.float32)))
    # the same oddities on a template that stays in global memory (k_icp_pipe_big in pipe mode)
    from perception_amd import templates as T
    big = T.template_xyz32(length=0.2, width=0.1, height=0.075, density=0.002)
    cases.append(("big, 300 m away", big + off, ((pick(big, 800) - big.mean(0)) @ R.T + big.mean(0) - 0.002).astype(np.float32) + off))
    bdup = np.concatenate([big[::2], big[::2]])[rng.permutation(2 * len(big[::2]))]
    cases.append(("big, duplicated points", bdup, (pick(big, 800) @ R.T + np.float32(0.003)).astype(np.float32)))
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "sliced", "cluster", "pipe"])
def test_icp_hard_geometry_in_every_driver(O, template, mode, monkeypatch):
    """Coordinates beyond the fast fixed-point range, a one-cell-thick grid, a template smaller than a run, exact distance
    ties between different original indices, sources a metre away - LDS-resident and global-memory templates, every driver:
    iterations, transform bits and fitness equal the oracle's."""
    monkeypatch.setenv("CUBOID_ICP_MODE", mode)
    ctx = capi.Context(max_points=8192, max_frames=1)
    try:
        for slot, (name, tpl, src) in enumerate(_hard_geometry_cases(template)):
            tpl = np.ascontiguousarray(tpl, np.float32)
            ctx.set_template(slot % capi.CD_MAX_TEMPLATES, tpl)
            prm = capi.default_params()
            prm.icp_max_iterations = 50
            st, res, al = ctx.icp(slot % capi.CD_MAX_TEMPLATES, src, prm, want_aligned=True)
            s0, r0, a0 = O.icp(tpl, src, prm, nn_mode=1, want_aligned=True)
            assert st == s0, (mode, name)
            assert (res.iterations, res.converged) == (r0.iterations, r0.converged), (mode, name)
            assert list(res.T) == list(r0.T) and res.fitness == r0.fitness, (mode, name)
            assert np.array_equal(al.view(np.uint32), a0.view(np.uint32)), (mode, name)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("max_wg", ["2", ""])
def test_big_template_batches_in_flight(max_wg, monkeypatch):
    """k_icp_pipe_big with other calls on the device (the regime in which a whole-cluster launch keeps four clusters in flight
    per workgroup): four batches of a 10 700-point template on two contexts give the records of a serial pass byte for byte -
    with the grid capped at two workgroups (every slot refills many times) and uncapped."""
    import torch
    from perception_amd import batch
    monkeypatch.setenv("CUBOID_ICP_MODE", "pipe")
    monkeypatch.setenv("CUBOID_ICP_SLOTS", "4")      # (the regime rule asks for four calls in flight; two contexts here)
    if max_wg:
        monkeypatch.setenv("CUBOID_ICP_MAX_WG", max_wg)
    big = templates.template_xyz32(length=0.2, width=0.1, height=0.075, density=0.002)
    sets = [np.stack([synth.frame(i) for i in idx], 0) for idx in ((0, 1, 2, 3), (4, 5, 6, 7), (8, 9, 10, 11), (12, 13, 14, 15))]
    prm = capi.default_params()
    N = sets[0].shape[1]
    ctx = capi.Context(max_points=N, max_frames=4)
    ctx.set_template(0, big)
    want = [capi.results_to_array(ctx.process_batch(s, prm)[0]).copy() for s in sets]
    ctx.close()
    dev = [torch.from_numpy(s).cuda() for s in sets]
    torch.cuda.synchronize()
    pipe = batch.BatchPipeline(N, 4, {0: big}, inflight=2)
    try:
        futs = [pipe.submit(d.data_ptr(), 16, N, 4, prm) for d in dev]
        for f, w in zip(futs, want):
            rec, _ = f.result()
            assert np.array_equal(rec, w)
    finally:
        pipe.close()
