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


@pytest.mark.parametrize("bad", ["float64", "offset", "step"])
def test_gps_node_refuses_layouts_the_device_code_cannot_read(bad, tmp_path):
    """x,y,z as FLOAT32 at byte offsets 0/4/8 in records of whole 4-byte words is what cd_ground_plane reads and writes; a
    PointCloud2 that says otherwise (a FLOAT64 field, z at another offset, a 30-byte point_step) gets ROS_ERROR and no
    publication - never a cloud of reinterpreted bytes under the input's field table."""
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    fin, fout = str(tmp_path / "frame.bin"), str(tmp_path / "out.bin")
    synth.frame(2).astype(np.float32).tofile(fin)
    r = subprocess.run([os.path.join(CPP, "gps_shim_driver"), fin, fout, "0.005", "0.015"], env=dict(os.environ, GPS_DRIVE_BAD=bad),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "nothing published" in r.stderr and not os.path.exists(fout)


def _read_cloud_dump(path):
    """a PointCloud2 as tests/ros_stubs/drive_opd.cpp dumps it"""
    blob = open(path, "rb").read()
    point_step, width, height, nf, dense, row_step = struct.unpack_from("<6i", blob, 0)
    pos = 24
    fields = []
    for _ in range(nf):
        off, dt, cnt = struct.unpack_from("<3i", blob, pos)
        fields.append((blob[pos + 12:pos + 28].split(b"\0")[0].decode(), off, dt, cnt))
        pos += 28
    frame = blob[pos:pos + 64].split(b"\0")[0].decode()
    data = np.frombuffer(blob, np.uint8, -1, pos + 64)
    return dict(point_step=point_step, width=width, height=height, dense=dense, row_step=row_step, fields=fields, frame=frame,
                data=data.reshape(width, point_step) if width else data.reshape(0, max(point_step, 1)))


XYZ16_FIELDS = [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1)]      # pcl::toROSMsg(PointCloud<PointXYZ>)


def test_gps_and_icp_nodes_chained(O, template, tmp_path):
    """iterative_closest_point.launch: gps publishes, icp subscribes.  Both node shims (separate objects, linked against the
    roscpp stand-ins) run back to back on one frame.  Checked against the reference's interface and the oracle:
    every subscription and publisher the two mains register (gps.cpp:146-150, icp.cpp:226-233); /icp/pose, /icp/bbox_points
    and the TF for the reference's chain (icp.cpp:150-182 on the whole extracted cloud, publish_pose / publish_bounding_box
    icp.cpp:55-128); the XYZ clouds in pcl::toROSMsg's 16-byte layout with the oracle's aligned points and the template's
    own points in them (icp.cpp:193-194); and the latch: later frames re-publish all four messages and the TF, nothing is
    recomputed (icp.cpp:139-147)."""
    from perception_amd import templates
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    frame = synth.frame(0)
    fin, tpath, clouds = str(tmp_path / "frame.bin"), str(tmp_path / "template.pcd"), str(tmp_path / "clouds.bin")
    frame.astype(np.float32).tofile(fin)
    open(tpath, "wb").write(templates.template_pcd_bytes(**templates.DEFAULT_TEMPLATE))
    out = subprocess.run([os.path.join(CPP, "chain_shim_driver"), fin, tpath, "0.0004", "0", "0", "0", "0", "0", "0", "0", "1", clouds],
                         check=True, capture_output=True, text=True, timeout=120).stdout
    d = {ln.split()[0]: ln.split()[1:] for ln in out.strip().splitlines()}
    prm = capi.default_params()
    prm.rgb_offset = 12
    st, vox, rgb, _, _ = O.crop_voxel(frame, prm, want_rgb=True)
    s1, c1, inl, _ = O.segment_plane(vox, prm)
    keep = np.ones(len(vox), bool)
    keep[inl] = False
    src = vox[keep]
    s2, r, aligned = O.icp(template, src, prm, nn_mode=1, want_aligned=True)
    assert st == 0 and s1 == 0 and s2 == 0 and r.accepted == 1
    # the two nodes' subscriptions and publications, nothing more
    assert d["registered"] == ["gps_sub", "1", "gps_points", "1", "gps_coeff", "1", "icp_sub_points", "1", "icp_sub_pose", "1", "aligned", "1",
                               "bbox", "1", "template", "1", "pose", "1", "n_adv", "6", "n_sub", "3"]
    assert d["gps_points"] == [str(len(src)), "point_step", "32"]
    assert d["published"] == ["pose", "1", "bbox", "1", "aligned", "1", "template", "1", "tf", "1"]
    pose = np.array(r.pose).reshape(4, 4)
    pos, q = O.pose_to_position_quaternion(pose)
    got = [float.fromhex(t) for t in d["pose"] if t != "quat"]
    assert got == list(pos) + list(q)
    box = O.bbox_corners(pose, 0.2, 0.1, 0.03)
    assert len(d["bbox"]) == 32                                   # 8 records of 4 words
    got_box = np.array([float.fromhex(t) for t in d["bbox"]], np.float32).reshape(8, 4)
    assert np.array_equal(got_box[:, :3].view(np.uint32), box.view(np.uint32)) and (got_box[:, 3] == 1.0).all()
    assert d["aligned_points"] == [str(len(src))] and d["template_points"] == [str(len(template))]
    assert d["layout"] == ["aligned", "1", "template", "1", "bbox", "1", "frames", "camera_depth_optical_frame|camera_depth_optical_frame|camera_depth_optical_frame"]
    blob = np.fromfile(clouds, np.float32).reshape(-1, 4)
    assert np.array_equal(blob[:len(src), :3].view(np.uint32), aligned.view(np.uint32))
    assert np.array_equal(blob[len(src):, :3].view(np.uint32), template.view(np.uint32))
    assert d["tf"] == ["camera_depth_optical_frame", "->", "icp_cuboid_frame"]
    assert d["republished"] == ["pose", "2", "aligned", "2", "template", "2", "bbox", "2", "tf", "2", "same_aligned", "1"]


def test_icp_node_with_the_surface_pose_opt_in(O, template, tmp_path):
    """icp.cpp:165-167 as the opt-in it is here (private parameter use_surface_pose): the node waits for a pose on
    /surface_segmentation/pose, moves the template by it and registers the scene against the moved template."""
    from conftest import quat_to_matrix
    from perception_amd import templates
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    frame = synth.frame(0)
    fin, tpath = str(tmp_path / "frame.bin"), str(tmp_path / "template.pcd")
    frame.astype(np.float32).tofile(fin)
    open(tpath, "wb").write(templates.template_pcd_bytes(**templates.DEFAULT_TEMPLATE))
    prm = capi.default_params()
    prm.rgb_offset = 12
    st, vox, rgb, _, _ = O.crop_voxel(frame, prm, want_rgb=True)
    s1, c1, inl, _ = O.segment_plane(vox, prm)
    keep = np.ones(len(vox), bool)
    keep[inl] = False
    src = vox[keep]
    s2, r0, _ = O.icp(template, src, prm, nn_mode=1)
    # a surface pose: the true one, a little off (what surface_normal_estimation would deliver)
    pos, q = O.pose_to_position_quaternion(np.array(r0.pose).reshape(4, 4))
    pos = pos + np.array([0.004, -0.003, 0.002])
    args = [repr(float(v)) for v in list(pos) + list(q)]
    out = subprocess.run([os.path.join(CPP, "chain_shim_driver"), fin, tpath, "0.0004", "1"] + args, check=True, capture_output=True, text=True,
                         timeout=120).stdout
    d = {ln.split()[0]: ln.split()[1:] for ln in out.strip().splitlines()}
    assert d["before_pose"] == ["published", "0"]
    Rf = quat_to_matrix(*[float(a) for a in args[3:]]).astype(np.float32)
    tf_ = np.array([float(a) for a in args[:3]]).astype(np.float32)
    X = template
    moved = np.stack([((Rf[i, 0] * X[:, 0] + Rf[i, 1] * X[:, 1]) + Rf[i, 2] * X[:, 2]) + tf_[i] for i in range(3)], 1).astype(np.float32)
    s3, r, _ = O.icp(moved, src, prm, nn_mode=1)
    assert s3 == 0 and r.accepted == 1 and r.iterations < r0.iterations
    p1, q1 = O.pose_to_position_quaternion(np.array(r.pose).reshape(4, 4))
    assert [float.fromhex(t) for t in d["pose"] if t != "quat"] == list(p1) + list(q1)
    assert d["template_points"] == [str(len(template))]


def test_opd_node_service_and_cached_pose(O, tmp_path):
    """object_pose_detection: a frame arrives, `detect_objects` is called for the screwdriver (id 1: template of 1370 points,
    the closest cluster differs by 42 < 250: success, opd.cpp:416-429) and for the eraser (id 2: 2979 points, every cluster is
    more than 1000 points away: the pick finds nothing, failure).  Checked: every endpoint main() registers (opd.cpp:476-485);
    the <output> cloud of each service call - the oracle's object cloud in the INPUT's 32-byte layout with its colours
    (opd.cpp:338-343); after the success every new frame re-publishes /icp/registered_pcl (the oracle's aligned cluster,
    16-byte XYZ records), /icp/template, /icp/pose, the grasp marker and the TF (opd.cpp:257-267, 96-174); /icp/bbox_points
    is advertised and never published (opd.cpp:265 is commented out); after the failure nothing is re-published."""
    from conftest import GOLDEN
    from perception_amd import pcd
    subprocess.run(["make", "-C", CPP], check=True, stdout=subprocess.DEVNULL)
    frame = synth.frame(1)
    fin = str(tmp_path / "frame.bin")
    frame.astype(np.float32).tofile(fin)
    out = subprocess.run([os.path.join(CPP, "opd_shim_driver"), fin, GOLDEN + os.sep, "0.005", "0.01", str(tmp_path), "1", "2"], check=True,
                         capture_output=True, text=True, timeout=120).stdout
    lines = [ln.split() for ln in out.strip().splitlines()]
    prm = capi.default_params()
    prm.rgb_offset = 12
    prm.leaf_size = 0.005
    prm.plane_distance_threshold = 0.01
    prm.crop2_enable = 1
    prm.cluster_enable = 1
    tpl = pcd.read_xyz(os.path.join(GOLDEN, "screwdriver_ascii_tf.pcd")).astype(np.float32)
    o = O.process_frame(frame, prm, tpl, want_clouds=True)
    ro = o["result"]
    sizes = [ro.clusters[k].size for k in range(ro.n_clusters)]
    diffs = [abs(s - len(tpl)) for s in sizes]
    k = int(np.argmin(diffs))
    assert diffs[k] < 250
    c = ro.clusters[k]
    assert lines[0] == ["registered", "sub_input", "1", "service", "1", "output", "1", "registered_pcl", "1", "bbox", "1", "template", "1", "pose", "1",
                        "marker", "1", "n_adv", "6", "n_sub", "1"]
    assert lines[1] == ["service", "id", "1", "returned", "1", "success", "1", "output_published", "1"]
    assert lines[2][:7] == ["chosen", "size", str(c.size), "iterations", str(c.iterations), "accepted", str(c.accepted)]
    assert float.fromhex(lines[2][8]) == c.fitness
    assert [float.fromhex(t) for t in lines[3][1:]] == [float(v) for v in c.T]
    assert lines[4] == ["per_two_frames", "registered_pcl", "2", "template", "2", "pose", "2", "marker", "2", "bbox", "0", "tf", "2"]
    pos, q = O.pose_to_position_quaternion(np.array(c.pose).reshape(4, 4))
    assert [float.fromhex(t) for t in lines[5] if t not in ("pose", "quat")] == list(pos) + list(q)
    # the grasp marker (opd.cpp:96-136): a half-transparent red 2 x 2 x 15 cm cube at the pose; TF to object_frame (opd.cpp:172)
    assert lines[6][:11] == ["marker", "frame", "camera_depth_optical_frame", "ns", "grasp_pose", "id", "0", "type", "1", "action", "0"]
    assert lines[6][11:13] == ["pose_equal", "1"]
    assert [float.fromhex(t) for t in lines[6][14:17]] == [0.02, 0.02, 0.15]
    assert [float.fromhex(t) for t in lines[6][18:22]] == [1.0, 0.0, 0.0, 0.5]
    assert lines[7] == ["tf", "camera_depth_optical_frame", "->", "object_frame", "origin_equal", "1"]
    # <output> (opd.cpp:338-343): object cloud, input's field table, 32-byte records, colours carried, junk padding not leaked
    st, vox, rgb, _, _ = O.crop_voxel(frame, prm, want_rgb=True)
    keepv = np.ones(len(vox), bool)
    keepv[o["plane_inliers"]] = False
    keepv &= ~((vox[:, 2].astype(np.float64) > prm.crop2_z_max) | (vox[:, 2].astype(np.float64) < prm.crop2_z_min))
    for oid in (1, 2):
        oc = _read_cloud_dump(str(tmp_path / ("%d_output.bin" % oid)))
        assert (oc["point_step"], oc["height"], oc["dense"], oc["row_step"]) == (32, 1, 1, 32 * oc["width"])
        assert oc["fields"] == [("x", 0, 7, 1), ("y", 4, 7, 1), ("z", 8, 7, 1), ("rgb", 16, 7, 1)] and oc["frame"] == "camera_depth_optical_frame"
        assert oc["width"] == ro.n_objects == int(keepv.sum())
        assert np.array_equal(oc["data"][:, 0:12].copy().view(np.uint32), o["objects"].view(np.uint32))
        assert np.array_equal(oc["data"][:, 16:20].copy().view(np.uint32).ravel(), rgb[keepv])
        assert not oc["data"][:, 12:16].any() and not oc["data"][:, 20:32].any()
    # /icp/registered_pcl = output_pcls[argmin] (opd.cpp:259): the aligned cluster; /icp/template: the template as loaded
    src = o["objects"][o["labels"] == k]
    s2, r2, aligned = O.icp(tpl, src, prm, nn_mode=1, want_aligned=True)
    rc = _read_cloud_dump(str(tmp_path / "1_registered.bin"))
    assert (rc["point_step"], rc["height"], rc["dense"], rc["row_step"], rc["width"]) == (16, 1, 1, 16 * len(src), len(src))
    assert rc["fields"] == XYZ16_FIELDS and rc["frame"] == "camera_depth_optical_frame"
    recs = rc["data"].copy().view(np.float32).reshape(-1, 4)
    assert np.array_equal(recs[:, :3].view(np.uint32), aligned.view(np.uint32)) and (recs[:, 3] == 1.0).all()
    tc = _read_cloud_dump(str(tmp_path / "1_template.bin"))
    assert (tc["point_step"], tc["width"]) == (16, len(tpl)) and tc["fields"] == XYZ16_FIELDS and tc["frame"] == "camera_depth_optical_frame"
    assert np.array_equal(tc["data"].copy().view(np.float32).reshape(-1, 4)[:, :3].view(np.uint32), tpl.view(np.uint32))
    # eraser: no cluster within 1000 points of the template's size -> failure; the earlier success is gone (ICP_SUCCESS false)
    assert lines[8] == ["service", "id", "2", "returned", "0", "success", "0", "output_published", "1"]
    assert lines[9] == ["per_two_frames", "registered_pcl", "0", "template", "0", "pose", "0", "marker", "0", "bbox", "0", "tf", "0"]


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
