"""cd_lattice_detect (host-only part of cd_set_template): which templates are unions of axis-aligned lattices.

Reference: cuboid_detection/templates/make_cuboid.py:38-55 writes face(X, Y) at z = -H/2, face(X, Z) at y = -W/2,
face(Y, Z) at x = -L/2, meshgrid order (first axis fastest), one after the other; the committed six-face
template_cuboid_L200_W100_H75.pcd repeats the three faces at the far sides.  Anything else must be refused (the ICP then
keeps its generic searches), because the closed-form nearest neighbour of k_icp_lat.hip relies on that structure bit by bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from perception_amd import capi, pcd, templates

F32 = np.float32


def test_make_cuboid_templates_are_three_lattice_faces():
    for L, W, H, d in ((0.2, 0.1, 0.03, 0.002), (0.2, 0.075, 0.1, 0.005), (0.15, 0.15, 0.05, 0.002), (0.1, 0.1, 0.1, 0.002), (0.2, 0.1, 0.075, 0.002)):
        t = templates.template_xyz32(L, W, H, d)
        nx, ny, nz = len(np.unique(t[:, 0])), len(np.unique(t[:, 1])), len(np.unique(t[:, 2]))
        # (constant axis, fast axis, first index, points along the fast axis, along the slow axis)
        assert capi.lattice_detect(t) == [(2, 0, 0, nx, ny), (1, 0, nx * ny, nx, nz), (0, 1, nx * ny + nx * nz, ny, nz)], (L, W, H, d)


def test_reference_six_face_template_and_committed_three_face_files():
    t = pcd.read_xyz(os.path.join(GOLDEN, "template_cuboid_L200_W100_H75.pcd")).astype(F32)
    faces = capi.lattice_detect(t)
    assert [f[0] for f in faces] == [2, 1, 0, 2, 1, 0] and [f[2] for f in faces] == [0, 5000, 8800, 10700, 15700, 19500]
    assert sum(f[3] * f[4] for f in faces) == len(t) == 21400
    for name in ("template_cuboid_L200_W100_H75_3faces.pcd", "template_cuboid_L200_W75_H100_3faces.pcd"):
        assert len(capi.lattice_detect(pcd.read_xyz(os.path.join(GOLDEN, name)).astype(F32))) == 3


def test_everything_else_is_refused(template):
    t = template
    assert capi.lattice_detect(t[::-1].copy()) == []                       # descending rows
    assert capi.lattice_detect(t[:-1].copy()) == []                        # a row one point short
    assert capi.lattice_detect(np.concatenate([t, t[:1]])) == []           # a stray point after the last face
    for i in (0, 1, 99, 100, 4999, 5000, 7249):
        m = t.copy()
        m[i, i % 3] = np.nextafter(m[i, i % 3], F32(10))                   # one coordinate one ulp off
        assert capi.lattice_detect(m) == [], i
    m = t.copy(); m[[10, 11]] = m[[11, 10]]
    assert capi.lattice_detect(m) == []                                    # two neighbours swapped
    m = t.copy(); m[5000:6500, 0] = np.tile(np.sort(np.random.default_rng(0).uniform(-0.1, 0.1, 100)).astype(F32), 15)
    assert capi.lattice_detect(m) == []                                    # a face with its own (and non-uniform) x table
    nonuni = np.sort(np.random.default_rng(1).uniform(0, 1, 40)).astype(F32)
    xx, yy = np.meshgrid(nonuni, np.arange(10, dtype=F32))
    assert capi.lattice_detect(np.stack([xx.ravel(), yy.ravel(), np.zeros(400, F32)], 1)) == []   # a product, but not near-uniform
    for name in ("eraser_ascii.pcd", "clamp_ascii_tf.pcd", "marker_ascii.pcd", "screwdriver_ascii_tf.pcd"):
        assert capi.lattice_detect(pcd.read_xyz(os.path.join(GOLDEN, name)).astype(F32)) == [], name
    bad = t.copy(); bad[17, 2] = np.nan
    assert capi.lattice_detect(bad) == []
    many = np.concatenate([np.stack([xx.ravel() * 0 + np.tile(np.arange(40, dtype=F32), 10), yy.ravel(), np.full(400, z, F32)], 1) for z in range(7)])
    assert capi.lattice_detect(many) == []                                 # seven faces: more than a cuboid has


def test_single_plane_and_table_limits():
    xx, yy = np.meshgrid(np.arange(30, dtype=F32) * F32(0.01), np.arange(20, dtype=F32) * F32(0.01))
    one = np.stack([xx.ravel(), np.full(600, 0.5, F32), yy.ravel()], 1)
    assert capi.lattice_detect(one) == [(1, 0, 0, 30, 20)]                 # one face: the y axis has no table of its own
    xx, yy = np.meshgrid(np.arange(700, dtype=F32), np.arange(40, dtype=F32))
    assert capi.lattice_detect(np.stack([xx.ravel(), yy.ravel(), np.zeros(28000, F32)], 1)) == []   # 741 table entries: over the LDS table budget
