"""The ground_plane_segmentation node shim, driven without ROS (tests/ros_stubs): what it publishes on
/ground_plane_segmentation/points must be field-for-field what cuboid_detection/src/ground_plane_segmentation.cpp:96-112
publishes - ExtractIndices<PCLPointCloud2> + fromPCL keep the INPUT's field table and point_step - with one record per
voxel centroid that is not on the plane, and the refined coefficients on /ground_plane_segmentation/coefficients."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from perception_amd import capi, synth

pytestmark = pytest.mark.gpu
CPP = os.path.join(ROOT, "perception_amd", "cpp")


def test_gps_node_publishes_the_input_layout(O, tmp_path):
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    frame = synth.frame(2)
    fin, fout = str(tmp_path / "frame.bin"), str(tmp_path / "out.bin")
    frame.astype(np.float32).tofile(fin)
    subprocess.run([os.path.join(CPP, "gps_shim_driver"), fin, fout, "0.005", "0.015"], check=True, timeout=120)
    blob = open(fout, "rb").read()
    point_step, width, height, nf, dense, ncoef = struct.unpack_from("<6i", blob, 0)
    pos = 24
    fields = []
    for _ in range(nf):
        off, dt, cnt = struct.unpack_from("<3i", blob, pos)
        name = blob[pos + 12:pos + 28].split(b"\0")[0].decode()
        fields.append((name, off, dt, cnt))
        pos += 28
    coeff = np.frombuffer(blob, np.float32, ncoef, pos)
    pos += 4 * ncoef
    data = np.frombuffer(blob, np.uint8, -1, pos)
    # the D435 / pcl::PointXYZRGB wire layout of the input survives: same fields, same 32-byte records
    assert (point_step, height, dense) == (32, 1, 1)
    assert fields == [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("rgb", 16, 7, 1)]
    assert len(data) == width * 32
    rec = data.reshape(width, 32)
    # expected content: the oracle's voxel cloud minus its refined plane inliers (gps.cpp: no second crop, no clustering)
    prm = capi.default_params()
    prm.rgb_offset = 12
    st, vox, rgb, _, _ = O.crop_voxel(frame, prm, want_rgb=True)
    s1, c1, inl, _ = O.segment_plane(vox, prm)
    assert st == 0 and s1 == 0
    keep = np.ones(len(vox), bool)
    keep[inl] = False
    assert width == int(keep.sum())
    assert np.array_equal(rec[:, 0:12].copy().view(np.float32).reshape(-1, 3).view(np.uint32), vox[keep].view(np.uint32))
    assert np.array_equal(rec[:, 16:20].copy().view(np.uint32).ravel(), rgb[keep])
    assert not rec[:, 12:16].any() and not rec[:, 20:32].any()          # padding is zero, not the input's junk
    assert np.array_equal(coeff.view(np.uint32), c1.view(np.uint32))


def test_gps_and_icp_nodes_chained(O, template, tmp_path):
    """iterative_closest_point.launch: gps publishes, icp subscribes.  Both node shims (separate objects, linked against the
    roscpp stand-ins) run back to back on one frame: /icp/pose, /icp/bbox_points and the TF must be what the oracle gives
    for the reference's chain (icp.cpp:150-182 on the whole extracted cloud, publish_pose / publish_bounding_box
    icp.cpp:55-128), the aligned cloud and the template are published once, and the second frame only republishes the
    latched result (icp.cpp:139-147)."""
    from perception_amd import templates
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    frame = synth.frame(0)
    fin, tpath = str(tmp_path / "frame.bin"), str(tmp_path / "template.pcd")
    frame.astype(np.float32).tofile(fin)
    open(tpath, "wb").write(templates.template_pcd_bytes(**templates.DEFAULT_TEMPLATE))
    out = subprocess.run([os.path.join(CPP, "chain_shim_driver"), fin, tpath, "0.0004"], check=True, capture_output=True, text=True,
                         timeout=120).stdout
    d = {ln.split()[0]: ln.split()[1:] for ln in out.strip().splitlines()}
    prm = capi.default_params()
    prm.rgb_offset = 12
    st, vox, rgb, _, _ = O.crop_voxel(frame, prm, want_rgb=True)
    s1, c1, inl, _ = O.segment_plane(vox, prm)
    keep = np.ones(len(vox), bool)
    keep[inl] = False
    src = vox[keep]
    s2, r, _ = O.icp(template, src, prm, nn_mode=1)
    assert st == 0 and s1 == 0 and s2 == 0 and r.accepted == 1
    assert d["gps_points"] == [str(len(src)), "point_step", "32"]
    assert d["published"] == ["pose", "1", "bbox", "1", "aligned", "1", "template", "1", "tf", "1"]
    pose = np.array(r.pose).reshape(4, 4)
    pos, q = O.pose_to_position_quaternion(pose)
    got = [float.fromhex(t) for t in d["pose"] if t != "quat"]
    assert got == list(pos) + list(q)
    box = O.bbox_corners(pose, 0.2, 0.1, 0.03)
    assert [float.fromhex(t) for t in d["bbox"]] == [float(v) for v in box.ravel()]
    assert d["aligned_points"] == [str(len(src))] and d["template_points"] == [str(len(template))]
    assert d["tf"] == ["camera_depth_optical_frame", "->", "icp_cuboid_frame"]
    assert d["republished"] == ["1", "aligned_again", "1"]


def test_opd_node_service_and_cached_pose(O, tmp_path):
    """object_pose_detection: a frame arrives, `detect_objects` is called for the screwdriver (id 1: template of 1370 points,
    the closest cluster differs by 42 < 250: success, opd.cpp:416-429) and for the eraser (id 2: 2979 points, every cluster is
    more than 1000 points away: the pick finds nothing, failure); after the success every new frame republishes the cached
    pose (opd.cpp:257-267).  Chosen cluster, its transformation and the published pose against the oracle."""
    from conftest import GOLDEN
    from perception_amd import pcd
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    frame = synth.frame(1)
    fin = str(tmp_path / "frame.bin")
    frame.astype(np.float32).tofile(fin)
    out = subprocess.run([os.path.join(CPP, "opd_shim_driver"), fin, GOLDEN + os.sep, "0.005", "0.01", "1", "2"], check=True,
                         capture_output=True, text=True, timeout=120).stdout
    lines = [ln.split() for ln in out.strip().splitlines()]
    prm = capi.default_params()
    prm.rgb_offset = 12
    prm.leaf_size = 0.005
    prm.plane_distance_threshold = 0.01
    prm.crop2_enable = 1
    prm.cluster_enable = 1
    tpl = pcd.read_xyz(os.path.join(GOLDEN, "screwdriver_ascii_tf.pcd")).astype(np.float32)
    ro = O.process_frame(frame, prm, tpl)["result"]
    sizes = [ro.clusters[k].size for k in range(ro.n_clusters)]
    diffs = [abs(s - len(tpl)) for s in sizes]
    k = int(np.argmin(diffs))
    assert diffs[k] < 250
    c = ro.clusters[k]
    assert lines[0] == ["service", "id", "1", "returned", "1", "success", "1"]
    assert lines[1][:7] == ["chosen", "size", str(c.size), "iterations", str(c.iterations), "accepted", str(c.accepted)]
    assert float.fromhex(lines[1][8]) == c.fitness
    assert [float.fromhex(t) for t in lines[2][1:]] == [float(v) for v in c.T]
    assert lines[3] == ["poses_published", "1"]
    pos, q = O.pose_to_position_quaternion(np.array(c.pose).reshape(4, 4))
    assert [float.fromhex(t) for t in lines[4] if t not in ("pose", "quat")] == list(pos) + list(q)
    # eraser: no cluster within 1000 points of the template's size -> failure; the earlier success is gone (ICP_SUCCESS false)
    assert lines[5] == ["service", "id", "2", "returned", "0", "success", "0"]
    assert lines[6] == ["poses_published", "0"]


def test_bbox_filter_and_surface_normal_nodes(O, tmp_path):
    """bbox_filter: everything is rejected until a CameraInfo has arrived (bbox_filter.cpp:33-34), then the kept records
    are the input's own records (all fields) of the points whose projection falls strictly inside the rectangle (:30-51,
    :96-101).  surface_normal_estimation: silent until the table plane's coefficients have arrived (sne.cpp:170), then pose, TF
    and the three plane coefficient messages (:231-233) - all against the oracle."""
    from conftest import rot_xyz
    from test_oracle_kat import _corner_cloud
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    drv = os.path.join(CPP, "bbox_sne_shim_driver")
    # --- bbox_filter
    prm = capi.default_params()
    prm.rgb_offset = 12
    st, vox, rgb, _, _ = O.crop_voxel(synth.frame(3), prm, want_rgb=True)
    cloud = np.zeros((len(vox), 4), np.float32)
    cloud[:, :3] = vox
    cloud[:, 3] = rgb.view(np.float32)
    fin, fout = str(tmp_path / "cloud.bin"), str(tmp_path / "out.bin")
    cloud.tofile(fin)
    P = [615.0, 0.0, 320.0, 0.0, 0.0, 615.0, 240.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    rect = [250, 180, 420, 330]
    out = subprocess.run([drv, "bbox", fin, fout] + [repr(v) for v in P] + [str(v) for v in rect], check=True, capture_output=True,
                         text=True, timeout=120).stdout.strip().splitlines()
    idx = O.bbox_filter(vox, P, rect)
    assert 0 < len(idx) < len(vox)
    assert out[0].split() == ["before_camera_info", "width", "0"]
    assert out[1].split() == ["after", "width", str(len(idx)), "point_step", "16", "fields", "4", "publications", "2"]
    assert open(fout, "rb").read() == cloud[idx].tobytes()
    # --- surface_normal_estimation
    rng = np.random.default_rng(21)
    R = rot_xyz(0.35, -0.2, 0.6)
    pts = _corner_cloud(R, np.array([0.02, -0.03, 0.55]), rng).astype(np.float32)
    ax = R[:, 2].astype(np.float32)
    fin2 = str(tmp_path / "corner.bin")
    np.ascontiguousarray(pts[:, :3], np.float32).tofile(fin2)
    out = subprocess.run([drv, "sne", fin2, repr(float(ax[0])), repr(float(ax[1])), repr(float(ax[2])), "0.002"], check=True,
                         capture_output=True, text=True, timeout=120).stdout.strip().splitlines()
    prm = capi.default_params()
    prm.plane_distance_threshold = 0.002
    so, ro = O.surface_frame(np.ascontiguousarray(pts[:, :3], np.float32), ax, prm)
    assert so == 0
    assert out[0].split() == ["before_coefficients", "poses", "0"]
    assert out[1].split() == ["after", "poses", "1", "tf", "1"]
    pos, q = O.pose_to_position_quaternion(np.array(ro.Rt, np.float64).reshape(4, 4))
    assert [float.fromhex(t) for t in out[2].split() if t not in ("pose", "quat")] == list(pos) + list(q)
    for line, k in zip(out[3:6], (2, 1, 0)):                          # normal_x <- coeff[2], normal_y <- coeff[1], normal_z <- coeff[0]
        assert [float.fromhex(t) for t in line.split()[1:]] == [float(v) for v in ro.coeff[k]]
    assert out[6].split() == ["tf", "camera_depth_optical_frame", "->", "estimated_cuboid_frame"]
