"""Known-answer tests of the oracle's stages (analytic cases; PCL semantics of SURVEY 8a)."""
import numpy as np
import pytest

from conftest import rot_xyz
from perception_amd import capi


def _mt19937_py(seed, n):
    mt = [0] * 624
    mt[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
    out, idx = [], 624
    for _ in range(n):
        if idx >= 624:
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            idx = 0
        y = mt[idx]
        idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        out.append(y & 0xFFFFFFFF)
    return np.array(out, dtype=np.uint32)


def test_mt19937_known_answers(O):
    assert O.mt19937_stream(5489, 10000)[-1] == 4123659995      # ISO C++ [rand.predef]
    s = O.mt19937_stream(12345, 2000)                           # PCL's fixed SAC seed
    assert np.array_equal(s, _mt19937_py(12345, 2000))


def test_passthrough_limits_are_double(O):
    f = np.float32
    vals = np.array([0.2, np.nextafter(f(0.2), f(0)), np.nextafter(f(0.2), f(1)), -0.2,
                     np.nextafter(f(-0.2), f(0)), np.nan, np.inf, 0.0, -0.0], dtype=np.float32)
    pts = np.zeros((len(vals), 3), np.float32)
    pts[:, 0] = vals
    keep = O.passthrough(pts, 0, -0.2, 0.2)
    # float32(0.2) = 0.200000003 > 0.2 (double) -> rejected; float32(-0.2) < -0.2 -> rejected
    assert list(keep) == [1, 4, 7, 8]
    pts2 = np.ones((3, 3), np.float32)
    pts2[1, 1] = np.nan            # non-finite y drops the point even if the field is fine
    assert list(O.passthrough(pts2, 2, 0.0, 1.0)) == [0, 2]


def test_voxel_grid_order_and_centroid(O, prm):
    prm.leaf_size = 0.1
    pts = np.array([[0.05, 0.05, 0.55], [0.16, 0.05, 0.55], [0.04, 0.06, 0.56], [0.05, 0.15, 0.55],
                    [0.05, 0.05, 0.65], [0.5, 0.0, 0.5], [0.0, 0.0, 2.0], [np.nan, 0, 0.5]], np.float32)
    st, vox, _, nc, grid = O.crop_voxel(pts, prm)
    assert st == 0 and nc == 5
    assert list(grid[3:]) == [2, 2, 2]
    # ascending idx = x fastest, then y, then z
    c0 = (pts[0] + pts[2]) / np.float32(2)
    exp = np.array([c0, pts[1], pts[3], pts[4]], np.float32)
    assert np.array_equal(vox, exp)


def test_voxel_rgb_average(O, prm):
    prm.leaf_size = 0.1
    prm.rgb_offset = 12
    pts = np.zeros((3, 4), np.float32)
    pts[:, :3] = [[0.01, 0.01, 0.51], [0.02, 0.02, 0.52], [0.03, 0.01, 0.53]]
    pts[:, 3] = np.array([(10 << 16) | (20 << 8) | 30, (11 << 16) | (21 << 8) | 33, (13 << 16) | (25 << 8) | 30],
                         np.uint32).view(np.float32)
    st, vox, rgb, nc, _ = O.crop_voxel(pts, prm, want_rgb=True)
    assert st == 0 and len(vox) == 1
    assert rgb[0] == (11 << 16) | (22 << 8) | 31       # (34/3, 66/3, 93/3) truncated


def test_voxel_leaf_too_small(O, prm):
    prm.leaf_size = 1e-5
    pts = np.array([[-0.19, -5, 0.1], [0.19, 5, 0.89]], np.float32)
    st, *_ = O.crop_voxel(pts, prm)
    assert st == capi.CD_ERR_LEAF_TOO_SMALL


def _plane_cloud(n_in=3000, n_out=400, seed=1):
    rng = np.random.RandomState(seed)
    nrm = np.array([0.1, -0.7, -0.7])
    nrm /= np.linalg.norm(nrm)
    e1 = np.cross(nrm, [1, 0, 0.3])
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(nrm, e1)
    p0 = np.array([0, 0, 0.55])
    ab = rng.uniform(-0.3, 0.3, (n_in, 2))
    P = p0 + ab[:, :1] * e1 + ab[:, 1:] * e2
    h = rng.uniform(0.05, 0.3, n_out) * rng.choice([-1, 1], n_out)
    ab2 = rng.uniform(-0.3, 0.3, (n_out, 2))
    Q = p0 + ab2[:, :1] * e1 + ab2[:, 1:] * e2 + h[:, None] * nrm
    pts = np.concatenate([P, Q]).astype(np.float32)
    perm = rng.permutation(len(pts))
    truth = np.zeros(len(pts), bool)
    truth[:n_in] = True
    return pts[perm], truth[perm], nrm, -nrm @ p0


def test_plane_exact_inliers(O, prm):
    pts, truth, nrm, d = _plane_cloud()
    st, coeff, inl, iters = O.segment_plane(pts, prm)
    assert st == 0
    assert np.array_equal(inl, np.nonzero(truth)[0])
    s = np.sign(coeff[:3] @ nrm)
    assert np.abs(s * coeff[:3] - nrm).max() < 1e-5 and abs(s * coeff[3] - d) < 1e-5
    # adaptive stop: w = 0.88 -> k = ln(0.01)/ln(1-w^3) ~ 4: a handful of iterations, not 1000
    assert 1 <= iters <= 12
    assert abs(np.linalg.norm(coeff[:3]) - 1) < 1e-6


def test_plane_trace_and_adaptive_replay(O, prm):
    """Hypotheses are a fixed function of (seed, N): replaying PCL's k-logic over the traced
    counts reproduces the iteration count segment_plane reports."""
    pts, truth, *_ = _plane_cloud(seed=3)
    tri, mod, cnt = O.ransac_trace(pts, prm, 64)
    assert len(tri) == 64 and tri.min() >= 0 and tri.max() < len(pts)
    assert all(len(set(t)) == 3 for t in tri.tolist())
    st, coeff, inl, iters = O.segment_plane(pts, prm)
    k, best, it = 1.0, -2**31, 0
    for c in cnt:
        if not (it < k):
            break
        if c < 0:
            continue
        if c > best:
            best = c
            w = best / len(pts)
            p = min(max(1 - w ** 3, np.finfo(float).eps), 1 - np.finfo(float).eps)
            k = np.log(1 - 0.99) / np.log(p)
        it += 1
    assert it == iters


def test_plane_no_model(O, prm):
    st, coeff, inl, iters = O.segment_plane(np.zeros((2, 3), np.float32), prm)
    assert st == capi.CD_ERR_NO_MODEL and len(inl) == 0
    # Axis-aligned collinear points: PCL's ratio test sees 0/0 = NaN != x and lets the sample
    # through; the cross product is 0, normalize() gives NaN, nothing is ever an inlier, w = 0
    # keeps k huge and the loop runs to max_iterations + 1.  PCL reports success with NaN
    # coefficients and no inliers; so does the restatement.
    line = np.zeros((50, 3), np.float32)
    line[:, 0] = np.arange(50) * 0.01
    st, coeff, inl, iters = O.segment_plane(line, prm)
    assert st == 0 and len(inl) == 0 and np.isnan(coeff).all() and iters == 1001
    # Collinear along a diagonal: all three ratios are equal -> samples rejected -> no model
    diag = np.outer(np.arange(1, 51), [0.01, 0.02, 0.03]).astype(np.float32)
    st, coeff, inl, iters = O.segment_plane(diag, prm)
    assert len(inl) == 0


def test_eigen33_vs_numpy(O):
    rng = np.random.RandomState(0)
    for _ in range(20):
        A = rng.randn(3, 3)
        cov = (A @ np.diag([1.0, 0.3, 0.01]) @ A.T).astype(np.float32)
        v = O.eigen33_smallest(cov)
        w, V = np.linalg.eigh(cov.astype(np.float64))
        assert abs(abs(v @ V[:, 0]) - 1) < 2e-3


def test_svd3(O):
    rng = np.random.RandomState(1)
    mats = [rng.randn(3, 3).astype(np.float32) for _ in range(20)]
    mats += [np.zeros((3, 3), np.float32), np.diag([3, 2, 0]).astype(np.float32),
             np.outer([1, 2, 3], [0.5, -1, 2]).astype(np.float32)]
    for A in mats:
        U, S, V = O.svd3(A)
        assert np.abs(U @ np.diag(S) @ V.T - A).max() < 2e-5 * max(1, np.abs(A).max())
        assert np.abs(U @ U.T - np.eye(3)).max() < 1e-5 and np.abs(V @ V.T - np.eye(3)).max() < 1e-5
        assert S[0] >= S[1] >= S[2] >= 0
        assert np.abs(S - np.linalg.svd(A.astype(np.float64), compute_uv=False)).max() < 1e-5 * max(1, S[0])


def _blob(center, n, r, rng):
    return (np.asarray(center) + rng.uniform(-r, r, (n, 3))).astype(np.float32)


def test_cluster_strict_radius_and_size_window(O, prm):
    tol = np.float32(0.02)
    # chain along x with gaps just below / exactly at the tolerance (strict <)
    xs = np.cumsum([0, 0.019, 0.019, 0.02, 0.019], dtype=np.float64)
    chain = np.zeros((5, 3), np.float32)
    chain[:, 0] = xs
    prm.cluster_min_size, prm.cluster_max_size = 1, 100
    d2 = (chain[3, 0] - chain[2, 0]) ** 2
    lab, sizes, k = O.cluster(chain, prm, mode=0)
    if d2 < np.float32(0.02 * 0.02):      # float32 rounding decides; the predicate is the spec
        assert k == 1
    else:
        assert k == 2 and list(sizes) == [3, 2] and list(lab) == [0, 0, 0, 1, 1]
    rng = np.random.RandomState(5)
    prm.cluster_min_size, prm.cluster_max_size = 200, 25000
    a = _blob([0, 0, 0.5], 199, 0.01, rng)
    b = _blob([0.2, 0, 0.5], 200, 0.01, rng)
    c = _blob([0.4, 0, 0.5], 300, 0.01, rng)
    pts = np.concatenate([a, b, c])
    perm = rng.permutation(len(pts))
    src = np.concatenate([np.full(199, -1), np.full(200, 1), np.full(300, 0)])[perm]
    for mode in (0, 1):
        lab, sizes, k = O.cluster(pts[perm], prm, mode=mode)
        assert k == 2 and list(sizes) == [300, 200]
        assert np.array_equal(lab, src)


def test_cluster_max_size_drops_whole_component(O, prm):
    g = np.stack(np.meshgrid(np.arange(160), np.arange(157)), -1).reshape(-1, 2) * 0.01
    big = np.zeros((len(g), 3), np.float32)
    big[:, :2] = g
    assert len(big) == 25120
    prm.cluster_max_size = 25119
    lab, sizes, k = O.cluster(big, prm, mode=1)
    assert k == 0 and (lab == -1).all()
    prm.cluster_max_size = 25120
    lab, sizes, k = O.cluster(big, prm, mode=1)
    assert k == 1 and sizes[0] == 25120 and (lab == 0).all()


def test_cluster_tie_break_and_grid_equals_brute(O, prm):
    rng = np.random.RandomState(9)
    prm.cluster_min_size = 5
    pts = np.concatenate([_blob([0.3, 0, 0.5], 50, 0.012, rng), _blob([0, 0, 0.5], 50, 0.012, rng),
                          rng.uniform(-0.5, 0.5, (600, 3)).astype(np.float32) * [1, 1, 0.05]])
    l0, s0, k0 = O.cluster(pts, prm, mode=0)
    l1, s1, k1 = O.cluster(pts, prm, mode=1)
    assert k0 == k1 and np.array_equal(l0, l1) and np.array_equal(s0, s1)
    assert all(s0[i] >= s0[i + 1] for i in range(len(s0) - 1))
    # equal sizes: the component holding the smaller first index gets the smaller label
    for a in range(k0 - 1):
        if s0[a] == s0[a + 1]:
            assert np.nonzero(l0 == a)[0][0] < np.nonzero(l0 == a + 1)[0][0]


def test_nn_kdtree_equals_brute_with_ties(O, template):
    rng = np.random.RandomState(2)
    tgt = np.concatenate([template[::7], template[::7][:50]])      # duplicates -> exact ties
    q = np.concatenate([rng.uniform(-0.6, 0.6, (500, 3)), tgt[:100] + 0.001, tgt[-20:]]).astype(np.float32)
    i0, d0 = O.nn(tgt, q, mode=0)
    i1, d1 = O.nn(tgt, q, mode=1)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    assert (i0[-20:] < len(tgt) - 50).all()      # lowest index wins an exact tie


def test_icp_recovers_known_transform(O, prm, template):
    # Perturbation below half the template's 2 mm grid pitch: every nearest neighbour is the
    # true correspondent, so point-to-point ICP must land on T_known (a larger offset settles
    # in one of the grid's many local minima - PCL does too).
    R = rot_xyz(np.deg2rad(0.2), np.deg2rad(-0.15), np.deg2rad(0.25))
    t = np.array([0.0004, -0.0003, 0.0005])
    Tk = np.eye(4)
    Tk[:3, :3] = R
    Tk[:3, 3] = t
    src = (template[::3].astype(np.float64) @ R.T + t).astype(np.float32)    # pose == Tk
    prm.icp_euclidean_fitness_epsilon = 1e-9       # tight: run to the fixed point
    st, res, al = O.icp(template, src, prm, nn_mode=1, want_aligned=True)
    assert st == 0 and res.converged == 1
    pose = np.array(res.pose).reshape(4, 4)
    assert np.linalg.norm(pose - Tk) < 1e-4
    assert res.fitness < 1e-10
    T = np.array(res.T, np.float64).reshape(4, 4)
    assert np.abs(T @ pose - np.eye(4)).max() < 1e-6
    assert np.abs(al - template[::3]).max() < 1e-4


def test_icp_few_points(O, prm, template):
    st, res, _ = O.icp(template, template[:2], prm)
    assert st == capi.CD_ERR_FEW_CORRESPONDENCES and res.converged == 0
    assert list(res.T) == list(np.eye(4, dtype=np.float32).ravel())


def test_icp_real_cluster_self_registration(O, prm):
    """Real D435 cluster from the reference tree registered against a perturbed copy."""
    import os
    from conftest import GOLDEN
    from perception_amd import pcd
    X = pcd.read_xyz(os.path.join(GOLDEN, "eraser_ascii.pcd"))
    R = rot_xyz(np.deg2rad(0.1), np.deg2rad(0.15), np.deg2rad(-0.1))     # sub-pixel-pitch offset
    c = X.mean(0).astype(np.float64)
    src = ((X - c) @ R.T + c + [0.0003, -0.0002, 0.0002]).astype(np.float32)
    prm.icp_euclidean_fitness_epsilon = 1e-9
    st, res, _ = O.icp(X, src, prm, nn_mode=1)
    assert st == 0 and res.converged == 1 and res.fitness < 1e-10


def _within_bbox_np(pts, P, rect):
    """Independent restatement of bbox_filter.cpp:30-51: double accumulation, float store, float divide."""
    P = np.asarray(P, np.float64).reshape(3, 4)
    x, y, z = (pts[:, i].astype(np.float64) for i in range(3))
    with np.errstate(all="ignore"):
        uvw = [((((P[r, 0] * x) + (P[r, 1] * y)) + (P[r, 2] * z)) + P[r, 3]).astype(np.float32) for r in range(3)]
        u = uvw[0] / uvw[2]
        v = uvw[1] / uvw[2]
        keep = (np.float32(rect[0]) < u) & (u < np.float32(rect[2])) & (np.float32(rect[1]) < v) & (v < np.float32(rect[3]))
    return np.nonzero(keep)[0].astype(np.int32)


def test_bbox_filter_predicate(O):
    from perception_amd import synth
    P = [synth.FX, 0, synth.CX, 0, 0, synth.FY, synth.CY, 0, 0, 0, 1, 0]
    rect = [200, 150, 420, 330]
    rng = np.random.default_rng(11)
    pts = np.empty((20000, 3), np.float32)
    pts[:, 0] = rng.uniform(-0.4, 0.4, len(pts))
    pts[:, 1] = rng.uniform(-0.3, 0.3, len(pts))
    pts[:, 2] = rng.uniform(0.2, 0.9, len(pts))
    # points that project exactly onto the rectangle's edges (strict '<' drops them), z = 0 (u = inf/nan),
    # points behind the camera, NaNs
    z = np.float32(0.5)
    edge = np.array([[(200 - synth.CX) / synth.FX * 0.5, 0.0, z], [0.0, (330 - synth.CY) / synth.FY * 0.5, z]], np.float32)
    extra = np.array([[0, 0, 0], [0.1, 0.1, 0], [0.0, 0.0, -0.5], [np.nan, 0, 0.5], [0, 0, 0.5]], np.float32)
    pts = np.concatenate([pts, edge, extra])
    keep = O.bbox_filter(pts, P, rect)
    assert np.array_equal(keep, _within_bbox_np(pts, P, rect))
    assert 0 < len(keep) < len(pts)
    assert len(pts) - 1 in keep and len(pts) - 2 not in keep and len(pts) - 5 not in keep
    # integer-pixel projection matrix: a point that lands exactly on x1 is rejected, one ulp inside is kept
    Pi = [100, 0, 0, 0, 0, 100, 0, 0, 0, 0, 1, 0]
    q = np.array([[1.0, 1.5, 1.0], [np.nextafter(np.float32(1.0), np.float32(2.0)), 1.5, 1.0], [2.0, 1.5, 1.0]], np.float32)
    assert list(O.bbox_filter(q, Pi, [100, 100, 200, 200])) == [1]


def _corner_cloud(R, t, rng, n_top=1500, n_side_a=900, n_side_b=500, noise=0.0004):
    """Three faces of a 0.2 x 0.1 x 0.06 box seen from a corner, in a frame rotated by R and moved by t."""
    top = np.c_[rng.uniform(-0.1, 0.1, n_top), rng.uniform(-0.05, 0.05, n_top), np.full(n_top, 0.03)]
    sa = np.c_[rng.uniform(-0.1, 0.1, n_side_a), np.full(n_side_a, -0.05), rng.uniform(-0.03, 0.03, n_side_a)]
    sb = np.c_[np.full(n_side_b, -0.1), rng.uniform(-0.05, 0.05, n_side_b), rng.uniform(-0.03, 0.03, n_side_b)]
    pts = np.concatenate([top, sa, sb]) + rng.normal(0, noise, (n_top + n_side_a + n_side_b, 3))
    pts = pts[rng.permutation(len(pts))]
    return (pts @ R.T + t).astype(np.float32)


def test_axis_constrained_plane_models(O, prm):
    """SACMODEL_PERPENDICULAR_PLANE / SACMODEL_PARALLEL_PLANE (surface_normal_estimation.cpp:118-123): the constraint
    decides which face wins, not the inlier count alone."""
    rng = np.random.default_rng(5)
    R = rot_xyz(0.0, 0.0, 0.0)
    pts = _corner_cloud(R, np.array([0, 0, 0.5]), rng, n_top=600, n_side_a=1500, n_side_b=400)
    prm.plane_distance_threshold = 0.002
    st, c0, inl0, _ = O.segment_plane(pts, prm)                      # unconstrained: the biggest face (side a, normal +-y)
    assert st == 0 and abs(c0[1]) > 0.99 and 1450 < len(inl0) < 1650
    prm.plane_model = capi.CD_PLANE_PERPENDICULAR
    prm.plane_axis[0], prm.plane_axis[1], prm.plane_axis[2] = 0.0, 0.0, 1.0
    prm.plane_eps_angle = 0.1
    st, c1, inl1, _ = O.segment_plane(pts, prm)                      # normal parallel to z: the top face
    assert st == 0 and abs(c1[2]) > 0.99 and 580 < len(inl1) < 720   # the face + the rims of the two side faces
    prm.plane_model = capi.CD_PLANE_PARALLEL                         # normal perpendicular to z: side a again
    st, c2, inl2, _ = O.segment_plane(pts, prm)
    assert st == 0 and abs(c2[1]) > 0.99 and np.array_equal(inl2, inl0)
    # an axis no face satisfies: the best hypothesis scores 0, PCL still reports its model, with no inliers
    prm.plane_model = capi.CD_PLANE_PERPENDICULAR
    prm.plane_axis[0], prm.plane_axis[1], prm.plane_axis[2] = 0.577, 0.577, 0.577
    prm.plane_max_iterations = 50
    st, c3, inl3, it3 = O.segment_plane(pts, prm)
    assert st == 0 and len(inl3) == 0 and it3 == 51


def test_surface_frame_known_answer(O, prm):
    """surface_normal_estimation.cpp:167-234: the pose assembled from three constrained planes equals the box frame."""
    rng = np.random.default_rng(9)
    R = rot_xyz(0.35, -0.2, 0.6)
    t = np.array([0.02, -0.03, 0.55])
    pts = _corner_cloud(R, t, rng)
    prm.plane_distance_threshold = 0.002
    table_normal = (R @ np.array([0, 0, 1.0])).astype(np.float32)    # the table is parallel to the top face
    st, res = O.surface_frame(pts, table_normal, prm, invert=True)
    assert st == 0
    n_pts = list(res.n_points)
    assert n_pts == sorted(n_pts, reverse=True) and abs(n_pts[0] - 1500) < 60 and abs(n_pts[1] - 900) < 60 and abs(n_pts[2] - 500) < 60
    Rt = np.array(list(res.Rt), np.float64).reshape(4, 4)
    n0, n1, n2 = Rt[:3, 2], Rt[:3, 1], Rt[:3, 0]
    # columns: +-z (top, largest), +-y (side a), +-x (side b) of the box, right-handed per sne.cpp:207-210
    assert abs(abs(n0 @ R[:, 2]) - 1) < 1e-4 and abs(abs(n1 @ R[:, 1]) - 1) < 1e-4 and abs(abs(n2 @ R[:, 0]) - 1) < 1e-4
    assert n2 @ np.cross(n1, n0) > 0.999
    # origin: centroid of the top face moved along its normal to the height of side a's centroid (sne.cpp:213-214)
    m0 = R @ np.array([0, 0, 0.03]) + t
    m1 = R @ np.array([0, -0.05, 0.0]) + t
    want = m0 - (n0 @ (m0 - m1)) * n0
    assert np.linalg.norm(Rt[:3, 3] - want) < 2e-3
    assert list(Rt[3]) == [0, 0, 0, 1]
    # fewer than three planes: the third fit has nothing to find
    flat = pts[np.abs((pts - t) @ R[:, 2] - 0.03) < 0.001]
    st, _ = O.surface_frame(flat, table_normal, prm)
    assert st == capi.CD_ERR_NO_MODEL


def test_icp_initial_guess_known_answers(O, template):
    """pcl::Registration::align(output, guess) (opt-in here, cd_params.icp_use_guess; the reference's authors prepared it at
    icp.cpp:130-134,165-167): the identity as a guess is the plain align(output); a guess G moves the source once, so
    align(src, G) is align(G src) with final = T' G (same iterations, same fitness, aligned cloud identical); a guess close to
    the answer cuts the iteration count."""
    rng = np.random.RandomState(11)
    R = rot_xyz(np.deg2rad(4.0), np.deg2rad(-3.0), np.deg2rad(6.0))
    t = np.array([0.012, -0.02, 0.35])
    src = ((template[::4].astype(np.float64) + rng.normal(0, 2e-4, (len(template[::4]), 3))) @ R.T + t).astype(np.float32)
    prm = capi.default_params()
    s0, r0, a0 = O.icp(template, src, prm, nn_mode=1, want_aligned=True)
    assert s0 == 0 and r0.converged == 1
    # identity guess == no guess, bit for bit
    prm.icp_use_guess = capi.CD_GUESS_PARAMS
    prm.icp_guess[:] = list(np.eye(4, dtype=np.float32).ravel())
    s1, r1, a1 = O.icp(template, src, prm, nn_mode=1, want_aligned=True)
    assert (r1.iterations, r1.converged, r1.fitness, list(r1.T)) == (r0.iterations, r0.converged, r0.fitness, list(r0.T))
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))
    # a guess = the inverse of the (slightly wrong) known motion
    G = np.eye(4)
    Rg = rot_xyz(np.deg2rad(3.5), np.deg2rad(-2.5), np.deg2rad(5.0))
    G[:3, :3] = Rg.T
    G[:3, 3] = -Rg.T @ (t + [0.002, -0.001, 0.003])
    Gf = G.astype(np.float32)
    prm.icp_guess[:] = list(Gf.ravel())
    s2, r2, a2 = O.icp(template, src, prm, nn_mode=1, want_aligned=True)
    assert s2 == 0 and r2.converged == 1 and r2.iterations < r0.iterations
    truth = np.eye(4)
    truth[:3, :3], truth[:3, 3] = R, t
    assert np.linalg.norm(np.array(r2.pose).reshape(4, 4) - truth) < 2e-3          # pose = inverse of final, guess included
    # ... equals the registration of the pre-moved source, composed with the guess
    moved = np.stack([((Gf[i, 0] * src[:, 0] + Gf[i, 1] * src[:, 1]) + Gf[i, 2] * src[:, 2]) + Gf[i, 3] for i in range(3)], 1).astype(np.float32)
    prm.icp_use_guess = capi.CD_GUESS_NONE
    s3, r3, a3 = O.icp(template, moved, prm, nn_mode=1, want_aligned=True)
    assert (r3.iterations, r3.converged) == (r2.iterations, r2.converged)
    assert np.array_equal(a2.view(np.uint32), a3.view(np.uint32))
    T3 = np.array(r3.T, np.float32).reshape(4, 4)
    comp = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            comp[i, j] = ((T3[i, 0] * Gf[0, j] + T3[i, 1] * Gf[1, j]) + T3[i, 2] * Gf[2, j]) + T3[i, 3] * Gf[3, j]
    assert np.allclose(np.array(r2.T, np.float32).reshape(4, 4), comp, atol=2e-6)   # (the oracle folds the guess in step by step)
