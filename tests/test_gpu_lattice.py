"""The closed-form nearest neighbour for lattice templates (perception_amd/csrc/k_icp_lat.hip) and the ICP kernel built on it.

Every template cuboid_detection/templates/make_cuboid.py:38-55 writes - and the reference's committed 21 400-point six-face
template_cuboid_L200_W100_H75.pcd - is a union of axis-aligned lattices; cd_set_template detects that and the ICP
(iterative_closest_point.cpp:170-178) then finds correspondences without a search.  Tested here, through the C-ABI:
 * cd_template_nearest against a numpy brute force with the oracle's arithmetic ((dx*dx + dy*dy) + dz*dz in float32, ties ->
   lowest original index): index and distance bits, near / far / mid-cell / +-300 m along a face normal (a tie walk across a
   whole face);
 * the ICP on lattice templates against the oracle and, byte for byte, against the generic searches (CUBOID_ICP_LATTICE=0),
   for the launch shapes of k_icp_lat (clusters per workgroup x waves per cluster, slots refilled from the queue), single clusters and batches, one template and several, lattice and arbitrary
   templates mixed in one batch."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from perception_amd import capi, pcd, synth, templates

pytestmark = pytest.mark.gpu

F32 = np.float32


def _cuboid_templates():
    out = [("L200_W100_H30_d2", templates.template_xyz32(0.2, 0.1, 0.03, 0.002)),
           ("L200_W75_H100_d5", templates.template_xyz32(0.2, 0.075, 0.1, 0.005)),
           ("L150_W150_H50_d2", templates.template_xyz32(0.15, 0.15, 0.05, 0.002)),
           ("L100_W100_H100_d2", templates.template_xyz32(0.1, 0.1, 0.1, 0.002))]
    for name in ("template_cuboid_L200_W100_H75.pcd", "template_cuboid_L200_W100_H75_3faces.pcd"):
        out.append((name, pcd.read_xyz(os.path.join(GOLDEN, name)).astype(F32)))
    return out


def _brute(P, Q):
    """(index, d2) of the nearest point of P for every query, canonical float32 arithmetic, first minimum."""
    idx = np.empty(len(Q), np.int32)
    d2 = np.empty(len(Q), F32)
    for k, q in enumerate(Q):
        dx = (q[0] - P[:, 0]).astype(F32); dy = (q[1] - P[:, 1]).astype(F32); dz = (q[2] - P[:, 2]).astype(F32)
        d = ((dx * dx).astype(F32) + (dy * dy).astype(F32)).astype(F32)
        d = (d + (dz * dz).astype(F32)).astype(F32)
        idx[k] = int(np.argmin(d))
        d2[k] = d[idx[k]]
    return idx, d2


def _queries(P, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = P.min(0).astype(np.float64), P.max(0).astype(np.float64)
    Q = []
    for k in range(n):
        mode = k % 7
        if mode == 0:
            q = rng.uniform(lo - 0.01, hi + 0.01)
        elif mode == 1:
            q = rng.uniform(lo - 1.5, hi + 1.5)
        elif mode == 2:
            q = P[rng.integers(len(P))] + rng.normal(0, 1e-4, 3)
        elif mode == 3:   # far along one axis: whole rows / faces tie in float32
            q = P[rng.integers(len(P))].astype(np.float64).copy()
            q[rng.integers(3)] += rng.choice([-300.0, 300.0, -30.0, 30.0, 3.0])
        elif mode == 4:   # exactly between lattice lines
            step = float(P[1, 0] - P[0, 0])
            q = P[rng.integers(len(P))].astype(np.float64) + 0.5 * step * rng.integers(0, 2, 3)
        elif mode == 5:
            q = rng.uniform(-300, 300, 3)
        else:             # on a template point
            q = P[rng.integers(len(P))].astype(np.float64)
        Q.append(q)
    return np.asarray(Q, F32)


@pytest.mark.parametrize("which", range(6))
def test_lattice_nearest_equals_brute_force(which):
    name, P = _cuboid_templates()[which]
    Q = _queries(P, 1400 if len(P) < 10000 else 700, seed=which)
    ctx = capi.Context(max_points=8192, max_frames=1)
    try:
        ctx.set_template(0, P)
        nf = ctx.template_lattice_faces(0)
        assert nf == (6 if len(P) == 21400 else 3), (name, nf)
        idx, d2 = ctx.template_nearest(0, Q)
    finally:
        ctx.close()
    bi, bd = _brute(P, Q)
    assert not np.isnan(d2).any(), name
    assert np.array_equal(d2.view(np.uint32), bd.view(np.uint32)), name
    bad = np.nonzero(idx != bi)[0]
    assert len(bad) == 0, (name, Q[bad[:3]], idx[bad[:3]], bi[bad[:3]])


def test_arbitrary_templates_are_not_lattices(template):
    ctx = capi.Context(max_points=8192, max_frames=1)
    try:
        obj = pcd.read_xyz(os.path.join(GOLDEN, "eraser_ascii_tf.pcd")).astype(F32)
        moved = template.copy()
        moved[1234, 1] = np.nextafter(moved[1234, 1], F32(1))      # one point one ulp off its lattice line
        for slot, t in enumerate((obj, moved, template[::-1].copy(), template[:-1].copy())):
            ctx.set_template(slot, t)
            assert ctx.template_lattice_faces(slot) == 0, slot
        with pytest.raises(capi.CuboidError) as e:
            ctx.template_nearest(0, np.zeros((4, 3), F32))
        assert e.value.status == capi.CD_ERR_INVALID_ARG
        ctx.set_template(0, template)                                # a slot can change kind
        assert ctx.template_lattice_faces(0) == 3
    finally:
        ctx.close()


def _icp_cases(template):
    rng = np.random.default_rng(5)
    from conftest import rot_xyz
    R = rot_xyz(0.05, -0.08, 0.3).astype(F32)
    sub = template[rng.choice(len(template), 1500, replace=False)]
    return [("small offset", (sub @ R.T + F32([0.004, -0.003, 0.002])).astype(F32)),
            ("half a metre away", (sub @ R.T + F32([0.1, 0.05, 0.5])).astype(F32)),
            ("300 m up the face normal", (sub + F32([0, 0, 300.0])).astype(F32)),          # iteration 0: every neighbour is a tie walk
            ("300 m along x", (sub[:257] + F32([-300.0, 0, 0])).astype(F32)),
            ("three points", (sub[:3] + F32(0.001)).astype(F32)),
            ("beyond the fast fixed-point range", (sub[:700] + F32([0, 400.0, 0])).astype(F32))]


@pytest.mark.parametrize("shape", ["", "1,1", "1,4", "2,2", "4,1", "8,2"])
def test_lattice_icp_equals_the_oracle_and_the_generic_search(O, template, shape, monkeypatch):
    """cd_icp on one cluster: iterations, transform bits, fitness and the aligned cloud equal the oracle's and the generic
    search's, for launch shapes of k_icp_lat (clusters per workgroup, waves per cluster); 6-face template included."""
    big = pcd.read_xyz(os.path.join(GOLDEN, "template_cuboid_L200_W100_H75.pcd")).astype(F32)
    if shape:
        monkeypatch.setenv("CUBOID_LAT_SHAPE", shape)
    got = {}
    for lattice in ("1", "0"):
        monkeypatch.setenv("CUBOID_ICP_LATTICE", lattice)
        ctx = capi.Context(max_points=8192, max_frames=1)
        try:
            ctx.set_template(0, template)
            ctx.set_template(1, big)
            prm = capi.default_params()
            prm.icp_max_iterations = 60
            for slot, tpl in ((0, template), (1, big)):
                for name, src in _icp_cases(template):
                    if slot == 1:
                        src = (src + F32([0.1, 0.05, 0.0375])).astype(F32)   # (that template's origin is a corner)
                    st, res, al = ctx.icp(slot, src, prm, want_aligned=True)
                    got[(lattice, slot, name)] = (st, res.iterations, res.converged, list(res.T), res.fitness, al.copy())
                    if lattice == "1":
                        s0, r0, a0 = O.icp(tpl, src, prm, nn_mode=1, want_aligned=True)
                        assert st == s0, (slot, name)
                        assert (res.iterations, res.converged) == (r0.iterations, r0.converged), (slot, name)
                        assert list(res.T) == list(r0.T) and res.fitness == r0.fitness, (slot, name)
                        assert np.array_equal(al.view(np.uint32), a0.view(np.uint32)), (slot, name)
        finally:
            ctx.close()
    for (lattice, slot, name), v in got.items():
        if lattice == "1":
            w = got[("0", slot, name)]
            assert v[:5] == w[:5] and np.array_equal(v[5].view(np.uint32), w[5].view(np.uint32)), (slot, name)


@pytest.mark.parametrize("shape", ["", "1,4", "1,16", "4,1", "4,1,3", "8,1,2", "2,4,50"])
def test_lattice_batch_equals_generic_batch_and_reports_its_search(O, template, shape, monkeypatch):
    """A batch through the fused call: records byte-identical with the lattice search and without, cd_timing says which ran,
    one ICP launch; a sample of the frames against the oracle.  Shapes: clusters per workgroup, waves per cluster, clusters
    per slot (> 1: slots are refilled from the queue; 50: ONE workgroup works through the whole batch)."""
    if shape:
        monkeypatch.setenv("CUBOID_LAT_SHAPE", shape)
    idx = list(range(100, 124))
    frames = np.stack([synth.frame(i) for i in idx], 0)
    prm = capi.default_params()
    rec = {}
    for lattice in ("1", "0"):
        monkeypatch.setenv("CUBOID_ICP_LATTICE", lattice)
        ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
        try:
            ctx.set_template(0, template)
            for rep in range(2):
                res, _, _ = ctx.process_batch(frames, prm)
            t = ctx.timing()
            assert t.icp_search == (1 if lattice == "1" else 0)
            if lattice == "1":
                assert t.icp_kernel_launches == 1
            rec[lattice] = capi.results_to_array(res).copy()
            if lattice == "1":
                for f in (0, 7, 23):
                    ro = O.process_frame(frames[f], prm, template)["result"]
                    for k in range(min(ro.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
                        a, b = res[f].clusters[k], ro.clusters[k]
                        assert (a.size, a.iterations, a.converged, a.accepted) == (b.size, b.iterations, b.converged, b.accepted), (f, k)
                        assert list(a.T) == list(b.T) and a.fitness == b.fitness, (f, k)
        finally:
            ctx.close()
    assert np.array_equal(rec["1"], rec["0"])


@pytest.mark.parametrize("shape", ["", "4,1,2", "2,2,7"])
def test_lattice_and_arbitrary_templates_in_one_batch(O, template, shape, monkeypatch):
    """template_slot = -1 (every cluster against every template, opd.cpp:376-413): two lattice templates and one scanned object in
    one batch - the lattice clusters go to k_icp_lat (slots of one workgroup hold clusters of different templates, and a refill
    brings a slot a cluster of another template: its tables are restaged), the others to the generic drivers; best-fitness
    records byte-identical to the all-generic run, cd_timing.icp_search = 2."""
    if shape:
        monkeypatch.setenv("CUBOID_LAT_SHAPE", shape)
    obj = pcd.read_xyz(os.path.join(GOLDEN, "eraser_ascii_tf.pcd")).astype(F32)
    small = templates.template_xyz32(0.2, 0.075, 0.1, 0.005)
    idx = list(range(30, 38))
    frames = np.stack([synth.frame(i) for i in idx], 0)
    prm = capi.default_params()
    prm.template_slot = -1
    rec = {}
    for lattice in ("1", "0"):
        monkeypatch.setenv("CUBOID_ICP_LATTICE", lattice)
        ctx = capi.Context(max_points=frames.shape[1], max_frames=len(frames))
        try:
            ctx.set_template(0, template)
            ctx.set_template(1, obj)
            ctx.set_template(2, small)
            res, _, _ = ctx.process_batch(frames, prm)
            assert ctx.timing().icp_search == (2 if lattice == "1" else 0)
            rec[lattice] = capi.results_to_array(res).copy()
            if lattice == "1":
                per = [ctx.cluster_results(f) for f in range(len(frames))]
                assert {c.template_slot for fr in per for c in fr} <= {0, 1, 2}
        finally:
            ctx.close()
    assert np.array_equal(rec["1"], rec["0"])
