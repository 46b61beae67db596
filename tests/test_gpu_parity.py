"""GPU parity tests: every stage and the fused batch path of libcuboid_hip.so (called through
the C-ABI) against the CPU oracle on the same inputs, and against the committed goldens.
Bit-exact for voxel clouds, plane coefficients/indices, cluster labels, ICP iteration counts
and transforms; the north-star tolerance (pose Frobenius < 1e-4) is asserted as well."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rot_xyz
from perception_amd import capi, synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-4   # BASELINE.json north_star: ICP pose Frobenius error < 1e-4


@pytest.fixture(scope="module")
def ctx(template):
    c = capi.Context(max_points=synth.WIDTH * synth.HEIGHT, max_frames=4)
    c.set_template(0, template)
    yield c
    c.close()


def _same_cluster(a, b):
    assert (a.size, a.iterations, a.converged, a.accepted) == (b.size, b.iterations, b.converged, b.accepted)
    pa, pb = np.array(a.pose).reshape(4, 4), np.array(b.pose).reshape(4, 4)
    assert np.linalg.norm(pa - pb) < POSE_TOL
    assert list(a.T) == list(b.T), "final transformation not bit-identical"
    assert a.fitness == b.fitness


def test_crop_voxel_bit_exact(ctx, O, frames4):
    prm = capi.default_params()
    prm.rgb_offset = 12
    for leaf in (0.005, 0.01, 0.001):
        prm.leaf_size = leaf
        f = frames4[1]
        vox, rgb, nc = ctx.crop_voxel(f, prm, want_rgb=True)
        st, vo, ro, nco, _ = O.crop_voxel(f, prm, want_rgb=True)
        assert st == 0 and nc == nco and vox.shape == vo.shape
        assert np.array_equal(vox.view(np.uint32), vo.view(np.uint32))
        assert np.array_equal(rgb, ro)


def test_crop_single_pass_equals_two_pass_and_y_overflow(ctx, O, template, frames4, monkeypatch):
    """The crop reads the input once (k_crop_fused: chained scan + absolute coordinate fields, re-keyed by the first sort
    pass).  Same outputs as the two-pass crop on a batch; a frame whose y cells do not fit the bit field (|y| > 327 m at
    leaf 0.005) sends the batch back through the two-pass path and still matches the oracle; empty and tiny frames."""
    prm = capi.default_params()
    prm.rgb_offset = 12
    batch = np.stack(frames4, 0)
    res1, pi1, lb1 = ctx.process_batch(batch, prm, want_indices=True)
    monkeypatch.setenv("CUBOID_CROP_TWO_PASS", "1")
    c2 = capi.Context(max_points=synth.WIDTH * synth.HEIGHT, max_frames=4)
    monkeypatch.delenv("CUBOID_CROP_TWO_PASS")
    try:
        c2.set_template(0, template)
        res2, pi2, lb2 = c2.process_batch(batch, prm, want_indices=True)
        assert np.array_equal(capi.results_to_array(res1), capi.results_to_array(res2))
        assert np.array_equal(pi1, pi2) and np.array_equal(lb1, lb2)
        for leaf in (0.005, 0.001, 0.02):
            prm.leaf_size = leaf
            a, b = ctx.crop_voxel(frames4[2], prm, want_rgb=True), c2.crop_voxel(frames4[2], prm, want_rgb=True)
            assert a[2] == b[2] and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1])
    finally:
        c2.close()
    prm.leaf_size = 0.005
    far = frames4[0].copy()
    keep = np.flatnonzero((np.abs(far[:, 0]) < 0.2) & (far[:, 2] > 0) & (far[:, 2] < 0.9))
    far[keep[::97], 1] = np.float32(400.0)     # 80000 cells away: outside the 17-bit y field
    far[keep[5::101], 1] = np.float32(-300.0)
    vox, rgb, nc = ctx.crop_voxel(far, prm, want_rgb=True)
    st, vo, ro, nco, _ = O.crop_voxel(far, prm, want_rgb=True)
    assert st == 0 and nc == nco and np.array_equal(vox.view(np.uint32), vo.view(np.uint32)) and np.array_equal(rgb, ro)
    # ragged tile ends: sizes around the 2048-point tile
    for n in (1, 63, 2047, 2048, 2049, 6145):
        pts = frames4[3][1000:1000 + 40 * n:40][:n].copy()
        vox, rgb, nc = ctx.crop_voxel(pts, prm, want_rgb=True)
        st, vo, ro, nco, _ = O.crop_voxel(pts, prm, want_rgb=True)
        assert st == 0 and nc == nco and np.array_equal(vox.view(np.uint32), vo.view(np.uint32)), n


def test_crop_voxel_edge_cases(ctx, O):
    prm = capi.default_params()
    # all-NaN, single point, limits exactly on float boundaries, stride 12 (no rgb)
    v, _, nc = ctx.crop_voxel(np.full((10, 3), np.nan, np.float32), prm)
    assert len(v) == 0 and nc == 0
    one = np.array([[0.1, 0.0, 0.5]], np.float32)
    v, _, nc = ctx.crop_voxel(one, prm)
    assert nc == 1 and np.array_equal(v, one)
    f = np.float32
    xs = np.array([0.2, np.nextafter(f(0.2), f(0)), -0.2, np.nextafter(f(-0.2), f(0)), 0.0], np.float32)
    pts = np.stack([xs, np.zeros(5, np.float32), np.full(5, 0.9, np.float32)], 1)
    pts[4, 2] = np.nextafter(f(0.9), f(0))
    v, _, nc = ctx.crop_voxel(pts, prm)
    st, vo, _, nco, _ = O.crop_voxel(pts, prm)
    assert nc == nco and np.array_equal(v, vo)
    prm.leaf_size = 1e-5
    big = np.array([[-0.19, -5, 0.1], [0.19, 5, 0.89]], np.float32)
    with pytest.raises(capi.CuboidError) as e:
        ctx.crop_voxel(big, prm)
    assert e.value.status == capi.CD_ERR_LEAF_TOO_SMALL


def test_voxel_random_ragged_cloud(ctx, O):
    """Unorganized cloud, many points per voxel, 20-byte records."""
    rng = np.random.RandomState(3)
    n = 50000
    rec = np.zeros((n, 5), np.float32)
    rec[:, :3] = rng.uniform([-0.25, -0.3, -0.1], [0.25, 0.3, 1.0], (n, 3))
    rec[rng.rand(n) < 0.05, 1] = np.inf
    rec[:, 4] = rng.randint(0, 1 << 24, n).astype(np.uint32).view(np.float32)
    prm = capi.default_params()
    prm.leaf_size = 0.02
    prm.rgb_offset = 16
    vox, rgb, nc = ctx.crop_voxel(rec, prm, want_rgb=True)
    st, vo, ro, nco, _ = O.crop_voxel(rec, prm, want_rgb=True)
    assert nc == nco and np.array_equal(vox.view(np.uint32), vo.view(np.uint32)) and np.array_equal(rgb, ro)


def test_voxel_long_runs_and_a_voxel_of_70000_points(ctx, O):
    """The centroid kernel's corners: runs longer than the eight points it fetches up front (64 neighbours in one voxel),
    voxels of many runs (the same voxel visited again and again), and a voxel of more than 65 536 points, whose colour sums
    pass 2^24 - there PCL's float accumulation rounds, and the kernel's integer channel sums are replayed in float."""
    rng = np.random.RandomState(11)
    n = 120000
    rec = np.zeros((n, 4), np.float32)
    # 70 000 points inside ONE 5 mm voxel (cell [0.100, 0.105) x [0.000, 0.005) x [0.500, 0.505)), bright colours
    rec[:70000, :3] = rng.uniform([0.1002, 0.0002, 0.5002], [0.1048, 0.0048, 0.5048], (70000, 3))
    # red channel: values whose exact mean sits a few 1/70000 beside an integer while PCL's sequential float sum (rounded to
    # even above 2^24) lands on the other side of it - integer sums and float sums give different colours here
    fsum = lambda v: float(np.cumsum(v.astype(np.float32), dtype=np.float32)[-1])
    red = None
    for _ in range(20):
        cand = rng.randint(236, 256, 70000).astype(np.int64)
        d = fsum(cand) - int(cand.sum())
        if abs(d) < 8:
            continue
        delta = 70000 * int(round(cand.sum() / 70000)) - int(np.sign(d)) * int(abs(d) // 2) - int(cand.sum())
        step = 1 if delta > 0 else -1
        cand[np.nonzero((cand + step >= 1) & (cand + step <= 254))[0][:abs(delta)]] += step
        if int(np.float32(cand.sum()) / np.float32(70000)) != int(np.float32(fsum(cand)) / np.float32(70000)):
            red = cand
            break
    assert red is not None
    rec[:70000, 3] = ((red.astype(np.uint32) << 16) | rng.randint(0, 1 << 16, 70000).astype(np.uint32)).view(np.float32)
    # stretches of 64 / 20 / 9 neighbours per voxel, the voxels revisited in a scrambled order (many runs per voxel)
    cells = rng.randint(0, 40, 50000 // 10)
    pts = np.repeat(cells, 10)[:50000]
    rec[70000:, 0] = -0.15 + 0.005 * (pts % 8) + rng.uniform(0.0005, 0.0045, 50000)
    rec[70000:, 1] = 0.005 * (pts // 8) + rng.uniform(0.0005, 0.0045, 50000)
    rec[70000:, 2] = 0.7 + rng.uniform(0.0005, 0.0045, 50000)
    rec[70000:, 3] = rng.randint(0, 1 << 24, 50000).astype(np.uint32).view(np.float32)
    rec[70000:70200, :3] = rec[70000, :3]                 # one run of 64 + 64 + 64 + 8 identical neighbours
    prm = capi.default_params()
    prm.rgb_offset = 12
    vox, rgb, nc = ctx.crop_voxel(rec, prm, want_rgb=True)
    st, vo, ro, nco, _ = O.crop_voxel(rec, prm, want_rgb=True)
    assert st == 0 and nc == nco == n and vox.shape == vo.shape
    assert np.array_equal(vox.view(np.uint32), vo.view(np.uint32)) and np.array_equal(rgb, ro)


def test_segment_plane_bit_exact(ctx, O, frames4):
    prm = capi.default_params()
    for f in frames4[:2]:
        st, vo, _, _, _ = O.crop_voxel(f, prm)
        s1, c1, i1, it1 = ctx.segment_plane(vo, prm)
        s0, c0, i0, it0 = O.segment_plane(vo, prm)
        assert s1 == s0 == 0 and it1 == it0
        assert np.array_equal(c1.view(np.uint32), c0.view(np.uint32))
        assert np.array_equal(i1, i0)


def test_segment_plane_hard_cases(ctx, O):
    prm = capi.default_params()
    rng = np.random.RandomState(11)
    # no dominant plane: needs many hypotheses (several sampler rounds)
    clutter = rng.uniform(-0.3, 0.3, (4000, 3)).astype(np.float32)
    s1, c1, i1, it1 = ctx.segment_plane(clutter, prm)
    s0, c0, i0, it0 = O.segment_plane(clutter, prm)
    assert (s1, it1) == (s0, it0) and np.array_equal(i1, i0) and np.array_equal(c1.view(np.uint32), c0.view(np.uint32))
    # fewer than 3 points: no model
    s1, *_ = ctx.segment_plane(np.zeros((2, 3), np.float32), prm)
    assert s1 == capi.CD_ERR_NO_MODEL
    # axis-aligned collinear cloud: PCL runs max_iterations+1 hypotheses and returns NaN coefficients
    line = np.zeros((50, 3), np.float32)
    line[:, 0] = np.arange(50) * 0.01
    s1, c1, i1, it1 = ctx.segment_plane(line, prm)
    s0, c0, i0, it0 = O.segment_plane(line, prm)
    assert (s1, it1, len(i1)) == (s0, it0, len(i0)) == (0, 1001, 0) and np.isnan(c1).all()
    # optimize off
    prm.plane_optimize = 0
    pts = np.concatenate([np.c_[rng.uniform(-.2, .2, (2000, 2)), np.full(2000, 0.5)], rng.uniform(-.2, .6, (300, 3))]).astype(np.float32)
    s1, c1, i1, it1 = ctx.segment_plane(pts, prm)
    s0, c0, i0, it0 = O.segment_plane(pts, prm)
    assert (s1, it1) == (s0, it0) and np.array_equal(i1, i0) and np.array_equal(c1.view(np.uint32), c0.view(np.uint32))


def _blob(center, n, r, rng):
    return (np.asarray(center) + rng.uniform(-r, r, (n, 3))).astype(np.float32)


def test_cluster_labels_bit_exact(ctx, O, frames4, template):
    prm = capi.default_params()
    r = O.process_frame(frames4[2], prm, template, want_clouds=True)
    lab, sizes, k = ctx.cluster(r["objects"], prm)
    assert k == r["result"].n_clusters and np.array_equal(lab, r["labels"])
    rng = np.random.RandomState(5)
    pts = np.concatenate([_blob([0, 0, .5], 199, .01, rng), _blob([.2, 0, .5], 200, .01, rng), _blob([.4, 0, .5], 300, .01, rng),
                          _blob([.6, 0, .5], 300, .01, rng), rng.uniform(-1, 1, (3000, 3)).astype(np.float32)])
    pts = pts[rng.permutation(len(pts))]
    for mn in (200, 1, 5):
        prm.cluster_min_size = mn
        lab, sizes, k = ctx.cluster(pts, prm, sizes_capacity=8192)
        l0, s0, k0 = O.cluster(pts, prm, mode=1, sizes_capacity=8192)
        assert k == k0 and np.array_equal(lab, l0) and np.array_equal(sizes, s0)
    # strict radius on a chain, max size drops the whole component
    chain = np.zeros((5, 3), np.float32)
    chain[:, 0] = np.cumsum([0, 0.019, 0.019, 0.02, 0.019])
    prm.cluster_min_size, prm.cluster_max_size = 1, 100
    lab, sizes, k = ctx.cluster(chain, prm)
    l0, s0, k0 = O.cluster(chain, prm, mode=0)
    assert k == k0 and np.array_equal(lab, l0)
    g = np.stack(np.meshgrid(np.arange(160), np.arange(157)), -1).reshape(-1, 2) * 0.01
    big = np.zeros((len(g), 3), np.float32)
    big[:, :2] = g
    prm.cluster_min_size, prm.cluster_max_size = 200, 25119
    lab, sizes, k = ctx.cluster(big, prm)
    assert k == 0 and (lab == -1).all()
    prm.cluster_max_size = 25120
    lab, sizes, k = ctx.cluster(big, prm)
    assert k == 1 and sizes[0] == 25120 and (lab == 0).all()


@pytest.mark.gpu
def test_cluster_cell_kernel_edges(ctx, O):
    """S5 in LDS clusters cells of edge tol/sqrt(3) (k_cluster.hip): exact-tie distances on a quantised lattice, dense
    slabs with many pairs near the radius, a cloud wider than its 10-bit cell coordinates and one with more cells than
    its table (both handed to the global-memory kernels) - labels and sizes identical to the oracle every time."""
    prm = capi.default_params()
    prm.cluster_min_size, prm.cluster_max_size = 1, 25000
    rng = np.random.RandomState(11)
    cases = {}
    # multiples of 2.5 mm: many pairs at exactly 0.02 (not connected, strict <) and at 0.0175 / 0.01903 (connected)
    q = rng.randint(0, 60, (5000, 3)).astype(np.float32) * np.float32(0.0025)
    cases["lattice"] = np.unique(q, axis=0)[:7000]
    slab = rng.uniform(0, 1, (6000, 3)).astype(np.float32) * np.float32([0.6, 0.6, 0.03])
    cases["slab"] = slab
    cases["sparse_slab"] = (rng.uniform(0, 1, (2500, 3)) * [1.2, 1.2, 0.02]).astype(np.float32)
    wide = np.concatenate([_blob([0, 0, .5], 400, .01, rng), _blob([13.0, 0, .5], 300, .01, rng),
                           _blob([0, -12.5, .5], 300, .01, rng)])
    cases["wide"] = wide[rng.permutation(len(wide))]
    cases["many_cells"] = np.concatenate([_blob([0, 0, .5], 500, .01, rng),
                                          rng.uniform(-2, 2, (5000, 3)).astype(np.float32)])
    cases["two_points"] = np.float32([[0, 0, 0], [0.0199, 0, 0]])
    cases["negative_coords"] = (slab[:3000] - np.float32([5.0, 7.0, 0.5])).astype(np.float32)
    for name, pts in cases.items():
        pts = np.ascontiguousarray(pts, np.float32)
        lab, sizes, k = ctx.cluster(pts, prm, sizes_capacity=8192)
        l0, s0, k0 = O.cluster(pts, prm, mode=0 if len(pts) <= 7000 else 1, sizes_capacity=8192)
        assert k == k0, (name, k, k0)
        assert np.array_equal(lab, l0), name
        assert np.array_equal(sizes, s0), name


@pytest.mark.gpu
@pytest.mark.parametrize("cells", ["1", "0"])
def test_cluster_beyond_the_lds_capacity(O, cells, monkeypatch):
    """Frames of more than 8192 object points (BASELINE config 5, the object launch values): the cell graph with the cells in
    LDS and the points in global memory (k_cluster_cells, end of round 5); CUBOID_CLUSTER_CELLS=0: the point-graph kernels of
    rounds 1-5.  Dense blobs and slabs (few cells, hundreds of points per cell), exact-tie distances, a cloud whose cells do
    not fit the table (falls through to the point-graph kernels either way), components beyond the size window - labels and
    sizes identical to the oracle."""
    monkeypatch.setenv("CUBOID_CLUSTER_CELLS", cells)
    cx = capi.Context(max_points=70000, max_frames=1)
    try:
        prm = capi.default_params()
        prm.cluster_min_size, prm.cluster_max_size = 1, 25000
        rng = np.random.RandomState(23)
        cases = {}
        cases["dense_blobs"] = np.concatenate([_blob([0, 0, .5], 9000, .03, rng), _blob([.3, 0, .5], 6000, .02, rng),
                                               _blob([.3, .2, .6], 3000, .005, rng), rng.uniform(-0.5, 0.5, (400, 3)).astype(np.float32)])
        cases["slab"] = (rng.uniform(0, 1, (20000, 3)) * [0.5, 0.4, 0.02]).astype(np.float32)
        q = rng.randint(0, 48, (40000, 3)).astype(np.float32) * np.float32(0.0025)     # exact ties at 0.02 (strict <)
        cases["lattice"] = np.unique(q, axis=0)[:30000]
        cases["too_many_cells"] = np.concatenate([_blob([0, 0, .5], 2000, .01, rng), rng.uniform(-1.5, 1.5, (9000, 3)).astype(np.float32)])
        cases["oversize_component"] = (rng.uniform(0, 1, (26000, 3)) * [0.3, 0.3, 0.01]).astype(np.float32)
        for name, pts in cases.items():
            pts = np.ascontiguousarray(pts[rng.permutation(len(pts))], np.float32)
            assert len(pts) > 8192
            lab, sizes, k = cx.cluster(pts, prm, sizes_capacity=16384)
            l0, s0, k0 = O.cluster(pts, prm, mode=1, sizes_capacity=16384)
            assert k == k0, (name, k, k0)
            assert np.array_equal(lab, l0), name
            assert np.array_equal(sizes, s0), name
        prm.cluster_min_size = 200      # the launch value: small components dropped
        lab, sizes, k = cx.cluster(cases["dense_blobs"], prm, sizes_capacity=16384)
        l0, s0, k0 = O.cluster(cases["dense_blobs"], prm, mode=1, sizes_capacity=16384)
        assert k == k0 and np.array_equal(lab, l0) and np.array_equal(sizes, s0)
    finally:
        cx.close()


def test_icp_bit_exact_and_known_answer(ctx, O, template, frames4):
    prm = capi.default_params()
    r = O.process_frame(frames4[0], prm, template, want_clouds=True)
    src = r["objects"][r["labels"] == 0]
    st, res, al = ctx.icp(0, src, prm, want_aligned=True)
    s0, r0, a0 = O.icp(template, src, prm, nn_mode=1, want_aligned=True)
    assert st == s0 == 0
    _same_cluster(res, r0)
    assert np.array_equal(al.view(np.uint32), a0.view(np.uint32))
    # known answer: sub-grid-pitch rigid offset of the template itself
    R = rot_xyz(np.deg2rad(0.2), np.deg2rad(-0.15), np.deg2rad(0.25))
    t = np.array([0.0004, -0.0003, 0.0005])
    Tk = np.eye(4)
    Tk[:3, :3], Tk[:3, 3] = R, t
    src = (template[::3].astype(np.float64) @ R.T + t).astype(np.float32)
    prm.icp_euclidean_fitness_epsilon = 1e-9
    st, res, _ = ctx.icp(0, src, prm)
    assert st == 0 and res.converged == 1
    assert np.linalg.norm(np.array(res.pose).reshape(4, 4) - Tk) < POSE_TOL
    s0, r0, _ = O.icp(template, src, prm, nn_mode=1)
    _same_cluster(res, r0)
    # too few points
    st, res, _ = ctx.icp(0, template[:2], prm)
    assert st == capi.CD_ERR_FEW_CORRESPONDENCES and res.converged == 0


def test_icp_large_template_and_real_cluster(ctx, O):
    """Template spanning many LDS chunks (21400 points, like the reference's 6-face template)
    and a real D435 cluster from the reference tree as the source."""
    from perception_amd import pcd, templates
    big = templates.template_xyz32(0.2, 0.1, 0.075, 0.001)[:21400]
    ctx.set_template(1, big)
    X = pcd.read_xyz(os.path.join(GOLDEN, "eraser_ascii.pcd"))
    prm = capi.default_params()
    src = (X - X.mean(0) + [0.01, 0.0, 0.0]).astype(np.float32)
    st, res, _ = ctx.icp(1, src, prm)
    s0, r0, _ = O.icp(big, src, prm, nn_mode=1)
    assert st == s0 == 0
    _same_cluster(res, r0)


def test_extract_indices_keeps_whole_records(ctx, O, frames4):
    """cd_extract = pcl::ExtractIndices<PCLPointCloud2> (gps.cpp:96-101): negative keeps the records NOT listed, in order;
    positive the listed ones in list order; 16-, 20- and 32-byte records come back field for field."""
    prm = capi.default_params()
    prm.rgb_offset = 12
    st, vox, rgb, _, _ = O.crop_voxel(frames4[0], prm, want_rgb=True)
    s1, c1, inl, _ = O.segment_plane(vox, prm)
    rng = np.random.RandomState(3)
    for words in (4, 5, 8):
        rec = rng.randint(0, 2 ** 31, (len(vox), words)).astype(np.uint32)
        rec[:, :3] = vox.view(np.uint32)
        got = ctx.extract(rec, inl, negative=True)
        keep = np.ones(len(vox), bool)
        keep[inl] = False
        assert np.array_equal(got, rec[keep])
        pick = rng.permutation(len(vox))[:777].astype(np.int32)
        assert np.array_equal(ctx.extract(rec, pick, negative=False), rec[pick])
    # edges: empty list, everything listed, duplicates and out-of-range entries in a negative list, ragged tile ends
    rec = np.arange(2049 * 4, dtype=np.uint32).reshape(2049, 4)
    assert np.array_equal(ctx.extract(rec, np.zeros(0, np.int32)), rec)
    assert len(ctx.extract(rec, np.arange(2049, dtype=np.int32))) == 0
    assert np.array_equal(ctx.extract(rec, np.array([5, 5, 2048, -3, 99999], np.int32)), np.delete(rec, [5, 2048], 0))
    with pytest.raises(capi.CuboidError):
        ctx.extract(rec, np.array([2049], np.int32), negative=False)


def test_persistent_sliced_kernel_equals_multi_launch(template, frames4, monkeypatch):
    """One frame / a few clusters: ONE persistent launch with a grid barrier per iteration (k_icp_persist) against the
    multi-launch loop (k_icp_solve + k_icp_iter per iteration, then k_icp_fitness): identical records."""
    prm = capi.default_params()
    prm.rgb_offset = 12
    out = {}
    for persist in ("1", "0", "2"):   # 2: the persistent launch gives up at its first barrier, the multi-launch loop takes over
        monkeypatch.setenv("CUBOID_ICP_PERSIST", persist)
        monkeypatch.setenv("CUBOID_ICP_MODE", "sliced")
        c = capi.Context(max_points=synth.WIDTH * synth.HEIGHT, max_frames=4)
        try:
            c.set_template(0, template)
            recs = []
            for f in range(4):
                r, _, _ = c.process_frame(frames4[f], prm)
                recs.append(bytes(r))
            rb, _, _ = c.process_batch(np.stack(frames4[:3], 0), prm)      # 5 clusters, still one launch
            out[persist] = (recs, bytes(capi.results_to_array(rb).tobytes()), c.timing().icp_kernel_launches)
        finally:
            c.close()
    assert out["1"][0] == out["0"][0] == out["2"][0] and out["1"][1] == out["0"][1] == out["2"][1]
    assert out["1"][2] == 1 and out["0"][2] > 10 and out["2"][2] > 10


def test_process_frame_is_a_batch_of_one(ctx, frames4):
    prm = capi.default_params()
    prm.rgb_offset = 12
    r1, pi1, lb1 = ctx.process_frame(frames4[1], prm, want_indices=True)
    rb, pib, lbb = ctx.process_batch(frames4[1][None], prm, want_indices=True)
    assert bytes(r1) == bytes(rb[0]) and np.array_equal(pi1, pib[0]) and np.array_equal(lb1, lbb[0])
    assert r1.n_clusters >= 1 and r1.status == 0


def test_batch_matches_oracle_and_goldens(ctx, O, template, frames4):
    prm = capi.default_params()
    prm.rgb_offset = 12
    batch = np.stack(frames4, 0)
    res, pi, lb = ctx.process_batch(batch, prm, want_indices=True)
    gold = json.load(open(os.path.join(GOLDEN, "frames_golden.json")))["frames"]
    for f in range(4):
        o = O.process_frame(frames4[f], prm, template, want_clouds=True)
        ro, rg, e = o["result"], res[f], gold[f]
        for k in ("status", "n_cropped", "n_voxels", "n_plane", "n_objects", "n_clusters", "ransac_iterations"):
            assert getattr(rg, k) == getattr(ro, k), (f, k)
        assert [float(x).hex() for x in rg.plane] == [float(x).hex() for x in ro.plane] == e["plane_hex"]
        assert np.array_equal(pi[f][:rg.n_plane], o["plane_inliers"]) and (pi[f][rg.n_plane:] == -1).all()
        assert np.array_equal(lb[f][:rg.n_objects], o["labels"]) and (lb[f][rg.n_objects:] == -1).all()
        assert hashlib.sha256(pi[f][:rg.n_plane].tobytes()).hexdigest() == e["plane_inliers_sha256"]
        assert hashlib.sha256(lb[f][:rg.n_objects].tobytes()).hexdigest() == e["labels_sha256"]
        for k in range(min(rg.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
            _same_cluster(rg.clusters[k], ro.clusters[k])
            assert [float(x).hex() for x in rg.clusters[k].T] == e["clusters"][k]["T_hex"]


def test_batch_cuboid_flavour_and_object_launch_params(ctx, O, template, frames4):
    """cluster_enable=0 (cuboid_detection: ICP on the whole extracted cloud) and the
    object_detection launch values (leaf 0.001, threshold 0.01)."""
    prm = capi.default_params()
    prm.cluster_enable = 0
    prm.crop2_enable = 0
    res, pi, lb = ctx.process_batch(np.stack(frames4[:2], 0), prm, want_indices=True)
    for f in range(2):
        o = O.process_frame(frames4[f], prm, template, want_clouds=True)
        assert res[f].n_clusters == o["result"].n_clusters == 1
        _same_cluster(res[f].clusters[0], o["result"].clusters[0])
    prm = capi.default_params()
    prm.leaf_size = 0.001
    prm.plane_distance_threshold = 0.01
    res, pi, lb = ctx.process_batch(frames4[3][None], prm, want_indices=True)
    o = O.process_frame(frames4[3], prm, template, want_clouds=True)
    rg, ro = res[0], o["result"]
    assert (rg.n_voxels, rg.n_plane, rg.n_objects, rg.n_clusters) == (ro.n_voxels, ro.n_plane, ro.n_objects, ro.n_clusters)
    assert np.array_equal(pi[0][:rg.n_plane], o["plane_inliers"]) and np.array_equal(lb[0][:rg.n_objects], o["labels"])
    for k in range(min(rg.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
        _same_cluster(rg.clusters[k], ro.clusters[k])


def test_constrained_plane_models_and_surface_frame_bit_exact(ctx, O, template, frames4):
    """Row 8f-3: SACMODEL_PERPENDICULAR_PLANE / PARALLEL_PLANE through cd_segment_plane and the whole
    surface_normal_estimation callback through cd_surface_frame, against the oracle: inlier indices, coefficients,
    iteration counts and the pose are bit-identical (the angular gate runs on the host in both, one libm)."""
    from test_oracle_kat import _corner_cloud
    from conftest import rot_xyz
    rng = np.random.default_rng(21)
    clouds = [_corner_cloud(rot_xyz(0.35, -0.2, 0.6), np.array([0.02, -0.03, 0.55]), rng),
              _corner_cloud(rot_xyz(-0.5, 0.3, -1.1), np.array([-0.05, 0.04, 0.6]), rng, n_top=700, n_side_a=1600, n_side_b=900)]
    axes = [rot_xyz(0.35, -0.2, 0.6)[:, 2], rot_xyz(-0.5, 0.3, -1.1)[:, 2]]
    for pts, ax in zip(clouds, axes):
        for model in (capi.CD_PLANE_PERPENDICULAR, capi.CD_PLANE_PARALLEL):
            for axis in (ax, np.array([0.577, 0.577, 0.577])):
                prm = capi.default_params()
                prm.plane_distance_threshold = 0.002
                prm.plane_model = model
                prm.plane_eps_angle = 0.1
                prm.plane_max_iterations = 200
                for i in range(3):
                    prm.plane_axis[i] = float(axis[i])
                sg, cg, ig, itg = ctx.segment_plane(pts, prm)
                so, co, io, ito = O.segment_plane(pts, prm)
                assert sg == so and itg == ito, (model, axis)
                assert [float(x).hex() for x in cg] == [float(x).hex() for x in co]
                assert np.array_equal(ig, io)
        prm = capi.default_params()
        prm.plane_distance_threshold = 0.002
        sg, rg = ctx.surface_frame(pts, ax.astype(np.float32), prm)
        so, ro = O.surface_frame(pts, ax.astype(np.float32), prm)
        assert sg == so == 0
        assert bytes(rg) == bytes(ro)
    # the real thing: the extracted object cloud of a synthetic frame and its table normal (gps -> sne in the reference)
    prm = capi.default_params()
    o = O.process_frame(frames4[1], prm, template, want_clouds=True)
    normal = np.array(list(o["result"].plane)[:3], np.float32)
    prm.plane_distance_threshold = 0.004
    sg, rg = ctx.surface_frame(o["objects"], normal, prm)
    so, ro = O.surface_frame(o["objects"], normal, prm)
    assert sg == so and bytes(rg) == bytes(ro)


def test_bbox_filter_stage_bit_exact(ctx, O, frames4):
    """cd_bbox_filter (stage level, what a bbox_filter node calls) against the oracle's within_bbox."""
    from perception_amd import synth
    P = [synth.FX, 0, synth.CX, 0, 0, synth.FY, synth.CY, 0, 0, 0, 1, 0]
    pts = frames4[0][:, :3].copy()
    pts[5] = [0, 0, 0]             # w = 0: u, v = nan -> dropped
    pts[6] = [0.1, 0.1, -0.5]      # behind the camera
    for rect in ([200, 150, 420, 330], [0, 0, 640, 480], [10, 10, 11, 11]):
        assert np.array_equal(ctx.bbox_filter(pts, P, rect), O.bbox_filter(pts, P, rect)), rect
    Pi = [100, 0, 0, 0, 0, 100, 0, 0, 0, 0, 1, 0]   # exact edge: strict '<'
    q = np.array([[1.0, 1.5, 1.0], [np.nextafter(np.float32(1.0), np.float32(2.0)), 1.5, 1.0], [2.0, 1.5, 1.0]], np.float32)
    assert list(ctx.bbox_filter(q, Pi, [100, 100, 200, 200])) == [1]
    assert len(ctx.bbox_filter(np.zeros((0, 3), np.float32), P, [0, 0, 1, 1])) == 0


def test_batch_with_bbox_filter_gate(ctx, O, template, frames4):
    """bbox_filter.cpp's image-space rectangle (row 8f-4) applied to the extracted cloud: object points,
    cluster labels and ICP results follow the oracle bit for bit, and the gate really removes points."""
    from perception_amd import synth
    prm = capi.default_params()
    P = [synth.FX, 0, synth.CX, 0, 0, synth.FY, synth.CY, 0, 0, 0, 1, 0]
    for i, v in enumerate(P):
        prm.bbox_P[i] = v
    base, _, _ = ctx.process_batch(np.stack(frames4, 0), prm, want_indices=True)
    prm.bbox_enable = 1
    for rect in ([250, 170, 400, 320], [0, 0, 640, 480], [300, 200, 301, 201]):
        for i, v in enumerate(rect):
            prm.bbox_rect[i] = v
        res, pi, lb = ctx.process_batch(np.stack(frames4, 0), prm, want_indices=True)
        for f in range(4):
            o = O.process_frame(frames4[f], prm, template, want_clouds=True)
            rg, ro = res[f], o["result"]
            for k in ("status", "n_voxels", "n_plane", "n_objects", "n_clusters"):
                assert getattr(rg, k) == getattr(ro, k), (rect, f, k)
            assert np.array_equal(pi[f][:rg.n_plane], o["plane_inliers"])
            assert np.array_equal(lb[f][:rg.n_objects], o["labels"])
            for k in range(min(rg.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
                _same_cluster(rg.clusters[k], ro.clusters[k])
            assert rg.n_plane == base[f].n_plane            # the gate acts on the extracted cloud only
            assert rg.n_objects <= base[f].n_objects
        if rect[0] == 250:
            assert any(0 < res[f].n_objects < base[f].n_objects for f in range(4))
        if rect[0] == 300:
            assert all(res[f].n_objects < 20 for f in range(4))


def test_batch_is_idempotent_and_frame_independent(ctx, frames4):
    """Size-independent properties: running twice gives identical bytes; a frame's record does
    not depend on its position in the batch or on its neighbours."""
    prm = capi.default_params()
    a, _, _ = ctx.process_batch(np.stack(frames4, 0), prm)
    b, _, _ = ctx.process_batch(np.stack(frames4[::-1], 0), prm)
    c, _, _ = ctx.process_batch(np.stack(frames4, 0), prm)
    A, B, Cc = capi.results_to_array(a), capi.results_to_array(b), capi.results_to_array(c)
    assert np.array_equal(A, Cc)
    assert np.array_equal(A, B[::-1])


def test_empty_and_degenerate_frames_in_batch(ctx, O, template, frames4):
    prm = capi.default_params()
    nanf = np.full_like(frames4[0], np.nan)
    planeonly = synth.render(dict(synth.scene_for(0), boxes=[]))
    res, _, _ = ctx.process_batch(np.stack([nanf, frames4[0], planeonly], 0), prm)
    assert res[0].n_cropped == 0 and res[0].n_voxels == 0 and res[0].status == capi.CD_ERR_NO_MODEL and res[0].n_clusters == 0
    o1 = O.process_frame(frames4[0], prm, template)["result"]
    _same_cluster(res[1].clusters[0], o1.clusters[0])
    o2 = O.process_frame(planeonly, prm, template)["result"]
    assert (res[2].n_voxels, res[2].n_plane, res[2].n_objects, res[2].n_clusters) == (o2.n_voxels, o2.n_plane, o2.n_objects, o2.n_clusters)


def test_icp_exact_ties_pick_lowest_original_index(ctx, O):
    """Queries exactly equidistant from 2, 4 or 8 template points (lattice with dyadic coordinates,
    template order shuffled so original-index order is unrelated to the library's internal spatial
    tiling): the neighbour must be the lowest ORIGINAL index, as in the oracle's ascending scan."""
    rng = np.random.RandomState(17)
    g = np.arange(8) / 16.0
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    tpl = lat[rng.permutation(len(lat))]
    ctx.set_template(2, tpl)
    half = 1.0 / 32.0
    q = []
    for _ in range(400):
        base = g[rng.randint(0, 7, 3)]
        kind = rng.randint(0, 4)
        off = np.array([half, 0, 0]) if kind == 0 else np.array([half, half, 0]) if kind == 1 else \
            np.array([half, half, half]) if kind == 2 else rng.uniform(0, 1 / 16, 3)
        q.append(base + rng.permutation(off))
    q = np.array(q, np.float32)
    i0, d0 = O.nn(tpl, q, mode=0)
    i1, d1 = O.nn(tpl, q, mode=1)
    assert np.array_equal(i0, i1)
    # at least a third of the queries have an exact multi-way tie
    d_all = ((q[:, None, :] - tpl[None, :, :]) ** 2).sum(-1)
    assert ((d_all == d_all.min(1, keepdims=True)).sum(1) > 1).mean() > 0.3
    prm = capi.default_params()
    prm.icp_max_iterations = 1
    st, res, al = ctx.icp(2, q, prm, want_aligned=True)
    s0, r0, a0 = O.icp(tpl, q, prm, nn_mode=0, want_aligned=True)
    assert st == s0 == 0
    _same_cluster(res, r0)
    assert np.array_equal(al.view(np.uint32), a0.view(np.uint32))


def _random_scene(rng):
    """A random table + boxes + clutter scene as an (n, 4) float32 cloud (x, y, z, rgb bits) with random holes/NaNs."""
    n_plane = int(rng.integers(2000, 30000))
    tilt = rot_xyz(*(rng.uniform(-0.6, 0.6, 3)))
    z0 = rng.uniform(0.3, 0.8)
    uv = rng.uniform(-0.35, 0.35, (n_plane, 2))
    plane = np.c_[uv, np.zeros(n_plane)] @ tilt.T + [0, 0, z0]
    parts = [plane + rng.normal(0, rng.uniform(0.0002, 0.002), plane.shape)]
    for _ in range(int(rng.integers(0, 4))):
        c = np.r_[rng.uniform(-0.15, 0.15, 2), 0.0] @ tilt.T + [0, 0, z0] - tilt[:, 2] * rng.uniform(0.01, 0.05)
        ext = rng.uniform(0.02, 0.12, 3)
        m = int(rng.integers(300, 5000))
        box = rng.uniform(-1, 1, (m, 3)) * ext
        face = rng.integers(0, 3, m)
        box[np.arange(m), face] = ext[face] * rng.choice([-1, 1], m)
        parts.append(box @ rot_xyz(*(rng.uniform(-1, 1, 3))).T + c)
    parts.append(rng.uniform([-0.5, -0.5, -0.1], [0.5, 0.5, 1.2], (int(rng.integers(0, 800)), 3)))
    pts = np.concatenate(parts).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    bad = rng.random(len(pts)) < 0.01
    pts[bad] = np.nan
    out = np.zeros((len(pts), 4), np.float32)
    out[:, :3] = pts
    out[:, 3] = rng.integers(0, 1 << 24, len(pts)).astype(np.uint32).view(np.float32)
    return out


def test_randomised_scenes_and_parameters(O, template):
    """40 random scenes x random launch parameters through the whole chain: every integer output identical, every
    float output bit-identical to the oracle (plane, indices, labels, ICP transforms, fitness)."""
    from conftest import rot_xyz as _r  # noqa: F401  (used by _random_scene through the module global)
    rng = np.random.default_rng(20190409)
    ctx2 = capi.Context(max_points=40000, max_frames=1)
    try:
        ctx2.set_template(0, template)
        small = templates_small()
        ctx2.set_template(1, small)
        for case in range(40):
            cloud = _random_scene(rng)
            prm = capi.default_params()
            prm.rgb_offset = 12 if case % 2 else -1
            prm.leaf_size = float(rng.choice([0.004, 0.005, 0.0075, 0.01]))
            prm.plane_distance_threshold = float(rng.choice([0.005, 0.01, 0.015]))
            prm.plane_max_iterations = int(rng.choice([50, 1000]))
            prm.crop_x_min, prm.crop_x_max = float(rng.uniform(-0.4, -0.1)), float(rng.uniform(0.1, 0.4))
            prm.crop_z_max = float(rng.uniform(0.7, 1.0))
            prm.crop2_enable = int(rng.integers(0, 2))
            prm.cluster_enable = int(rng.integers(0, 2))
            prm.cluster_tolerance = float(rng.choice([0.01, 0.02, 0.03]))
            prm.cluster_min_size = int(rng.choice([20, 100, 200]))
            prm.icp_max_iterations = int(rng.choice([5, 40, 5000]))
            prm.template_slot = int(rng.integers(0, 2))
            tpl = template if prm.template_slot == 0 else small
            res, pi, lb = ctx2.process_batch(cloud[None], prm, want_indices=True)
            o = O.process_frame(cloud, prm, tpl, want_clouds=True)
            rg, ro = res[0], o["result"]
            for k in ("status", "n_cropped", "n_voxels", "n_plane", "n_objects", "n_clusters", "ransac_iterations"):
                assert getattr(rg, k) == getattr(ro, k), (case, k)
            assert [float(x).hex() for x in rg.plane] == [float(x).hex() for x in ro.plane], case
            assert np.array_equal(pi[0][:max(rg.n_plane, 0)], o["plane_inliers"]), case
            assert np.array_equal(lb[0][:max(rg.n_objects, 0)], o["labels"]), case
            for k in range(min(max(rg.n_clusters, 0), capi.CD_MAX_CLUSTERS_PER_FRAME)):
                _same_cluster(rg.clusters[k], ro.clusters[k])
    finally:
        ctx2.close()


def templates_small():
    from perception_amd import templates as T
    return T.template_xyz32(length=0.2, width=0.075, height=0.1, density=0.005)


@pytest.mark.parametrize("path", ["crop_runs", "crop_runs_copy", "crop_runs_quads", "runs", "points", "runs_off_for_big_contexts"])
def test_voxel_stage_by_runs_and_by_points(O, frames4, path, monkeypatch):
    """S1 sorts RUNS of equal voxel index among the cropped points.  Default since round 4 ("crop_runs"): the crop itself
    writes the runs and their digit histograms and the sort runs on the packed cell keys, only over the digits that vary
    (k_crop_runs, k_voxel_centroid_runs); CUBOID_CROP_RUNS=0 ("runs"): the crop writes per-point keys and k_voxel_runs finds the
    runs (rounds 2-3); CUBOID_VOXEL_RUNS=0 sorts the points (k_radix_ghist, k_voxel_centroid), and so does a context of more
    than 2^20 points per frame (a run's start has 20 bits).  Same bits as the oracle either way, on: organised frames (runs of
    ~2.4 points), one voxel holding 3000 consecutive points (runs are cut at every 64), keys that alternate from point to
    point (every run is one point long), an unorganised cloud with rgb, frames in a batch.
    End of round 5: 16-byte records are left in place (run records point into the input) and the centroids are summed one lane
    per voxel; "crop_runs_copy" (CUBOID_CROP_DIRECT=0) compacts the kept points as before, "crop_runs_quads"
    (CUBOID_CENTROID_LANES=0) is the quad-per-voxel kernel of rounds 3-5; a 32-byte record layout (below) always takes the copy."""
    if path == "crop_runs_copy":
        monkeypatch.setenv("CUBOID_CROP_DIRECT", "0")
    if path == "crop_runs_quads":
        monkeypatch.setenv("CUBOID_CENTROID_LANES", "0")
    if path == "points":
        monkeypatch.setenv("CUBOID_VOXEL_RUNS", "0")
    if path == "runs":
        monkeypatch.setenv("CUBOID_CROP_RUNS", "0")
    rng = np.random.RandomState(12)
    max_points = (1 << 20) + 4096 if path == "runs_off_for_big_contexts" else synth.WIDTH * synth.HEIGHT
    cx = capi.Context(max_points=max_points, max_frames=2)
    try:
        prm = capi.default_params()
        prm.rgb_offset = 12
        clouds = [frames4[0], frames4[2]]
        one = np.zeros((5000, 4), np.float32)
        one[:, :3] = [0.101, 0.051, 0.501]
        one[:3000, :3] += rng.uniform(0, 0.0039, (3000, 3)).astype(np.float32)       # 3000 consecutive points in one voxel
        one[3000:, :3] = rng.uniform([-0.19, -0.2, 0.1], [0.19, 0.2, 0.85], (2000, 3))
        one[:, 3] = rng.randint(0, 1 << 24, 5000).astype(np.uint32).view(np.float32)
        clouds.append(one)
        alt = np.zeros((4001, 4), np.float32)
        alt[0::2, :3] = [0.1, 0.0, 0.4]
        alt[1::2, :3] = [-0.1, 0.02, 0.6]
        alt[:, :3] += rng.uniform(0, 0.004, (4001, 3)).astype(np.float32)
        clouds.append(alt)
        for pts in clouds:
            vox, rgb, nc = cx.crop_voxel(pts, prm, want_rgb=True)
            st, vo, ro, nco, _ = O.crop_voxel(pts, prm, want_rgb=True)
            assert nc == nco and np.array_equal(vox.view(np.uint32), vo.view(np.uint32)) and np.array_equal(rgb, ro)
        if path.startswith("crop_runs"):
            # PointXYZRGB's layout (32-byte records, rgb at offset 16): never the in-place form
            wide = np.zeros((len(clouds[0]), 8), np.float32)
            wide[:, :3] = clouds[0][:, :3]
            wide[:, 4] = clouds[0][:, 3]
            prm32 = capi.default_params()
            prm32.rgb_offset = 16
            vox, rgb, nc = cx.crop_voxel(wide, prm32, want_rgb=True)
            st, vo, ro, nco, _ = O.crop_voxel(clouds[0], prm, want_rgb=True)
            assert nc == nco and np.array_equal(vox.view(np.uint32), vo.view(np.uint32)) and np.array_equal(rgb, ro)
        if path != "runs_off_for_big_contexts":
            from perception_amd import templates
            cx.set_template(0, templates.template_xyz32(**templates.DEFAULT_TEMPLATE))
            batch = np.stack([frames4[1], frames4[3]], 0)
            res, _, _ = cx.process_batch(batch, prm)
            want = [O.crop_voxel(batch[f], prm)[1] for f in range(2)]
            for f in range(2):
                got = cx.frame_cloud(f, capi.CD_CLOUD_VOXELS, 16, -1)[:, :3].copy().view(np.float32)
                assert res[f].n_voxels == len(want[f]) and np.array_equal(got.view(np.uint32), want[f].view(np.uint32))
    finally:
        cx.close()


def test_crop_runs_sort_passes_follow_the_digits_that_vary(ctx, O):
    """k_crop_runs' sort runs on the packed cell keys and only over the 8-bit digits that take more than one value in some
    frame (k_digit_vary).  The bench frames need three; here: a cloud whose y cells span more than an aligned block of 512
    (y in +-2 m at the 5 mm leaf: the digit above the ninth y bit varies - four passes), a cloud inside ONE voxel (no digit
    varies - no pass at all), and one that differs in the x cell only (one pass).  Bits equal to the oracle's every time."""
    rng = np.random.RandomState(21)
    prm = capi.default_params()
    prm.rgb_offset = 12

    def check(pts):
        vox, rgb, nc = ctx.crop_voxel(pts, prm, want_rgb=True)
        st, vo, ro, nco, _ = O.crop_voxel(pts, prm, want_rgb=True)
        assert st == 0 and nc == nco and len(vox) == len(vo)
        assert np.array_equal(vox.view(np.uint32), vo.view(np.uint32)) and np.array_equal(rgb, ro)
        return len(vo)

    wide = np.zeros((40000, 4), np.float32)
    wide[:, :3] = rng.uniform([-0.19, -2.0, 0.05], [0.19, 2.0, 0.85], (40000, 3))
    wide[:, 3] = rng.randint(0, 1 << 24, 40000).astype(np.uint32).view(np.float32)
    assert check(wide) > 30000
    one = np.zeros((777, 4), np.float32)
    one[:, :3] = np.float32([0.101, 0.051, 0.501]) + rng.uniform(0, 0.0039, (777, 3)).astype(np.float32)
    assert check(one) == 1
    xonly = one.copy()
    xonly[:, 0] += (np.arange(777) % 5).astype(np.float32) * np.float32(0.005)     # five neighbouring x cells, same y and z cell
    assert check(xonly) == 5
