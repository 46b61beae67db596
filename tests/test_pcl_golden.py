"""Row (c) of SURVEY.md section 8 - the oracle's parity with REAL PCL - is unpinned: the reference holds no golden output for
its point-cloud path and PCL cannot be built in this image.  What can be kept ready is the way to close it:

* tools/pcl_golden.cpp runs the exact PCL calls of object_detection/src/object_pose_detection.cpp:270-413 on the synthetic
  frames and writes tests/golden/pcl_frames_golden.json (schema of frames_golden.json).  Here it is only PARSED, against the
  declaration-only stand-ins of tests/pcl_stubs (as the ROS shims are against tests/ros_stubs) - that is not parity.
* when a maintainer with PCL has committed that file, the second test compares the oracle with it at north_star's
  tolerances (counts, plane indices and cluster labels exact; ICP pose Frobenius < 1e-4) and the row is pinned.  Until then it
  is skipped with the reason spelled out, and DESIGN.md says "parity unpinned"."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from perception_amd import capi, synth, templates

PCL_GOLDEN = os.path.join(GOLDEN, "pcl_frames_golden.json")
# the tool's flavours (tools/pcl_golden.cpp): file it writes, and the cd_params that restate the same reference code
FLAVOURS = {
    "chain": ("pcl_frames_golden.json", {}),
    "cuboid": ("pcl_frames_golden_cuboid.json", {"crop2_enable": 0, "cluster_enable": 0}),
    "object": ("pcl_frames_golden_object.json", {"leaf_size": 0.001, "plane_distance_threshold": 0.01}),
}


def test_pcl_golden_tool_parses_against_the_stand_ins():
    r = subprocess.run(["g++", "-std=c++14", "-Wall", "-fsyntax-only", "-I" + os.path.join(ROOT, "tests", "pcl_stubs"),
                        os.path.join(ROOT, "tools", "pcl_golden.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_frame_writer_for_the_pcl_tool(tmp_path):
    subprocess.run(["python", os.path.join(ROOT, "tools", "write_synth_frames.py"), str(tmp_path), "1"], check=True, stdout=subprocess.DEVNULL)
    f = np.fromfile(tmp_path / "frame_0.bin", "<f4").reshape(-1, 4)
    assert np.array_equal(f.view(np.uint32), synth.frame(0).view(np.uint32))
    want = json.load(open(os.path.join(GOLDEN, "frames_golden.json")))["frames"][0]["frame_sha256"]
    assert hashlib.sha256(f.tobytes()).hexdigest() == want          # the tool hashes the same bytes
    assert open(tmp_path / "template.pcd", "rb").read() == templates.template_pcd_bytes(**templates.DEFAULT_TEMPLATE)


def test_flavours_of_the_tool_are_the_parameter_sets_the_oracle_runs(O):
    """the tool's three flavours exist in its source, and the oracle runs each parameter set on frame 0 (so a golden file of any
    flavour can be compared the day it appears): the cuboid flavour has exactly one ICP source = the whole extracted cloud"""
    src = open(os.path.join(ROOT, "tools", "pcl_golden.cpp")).read()
    tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
    f = synth.frame(0)
    seen = {}
    for name, (_, over) in FLAVOURS.items():
        assert '{"%s", ' % name in src
        if name == "object":
            continue        # (leaf 0.001: 110 k voxels per frame - the GPU suite and bench.py's object_launch leg run it against the oracle)
        prm = capi.default_params()
        prm.rgb_offset = 12
        for k, v in over.items():
            setattr(prm, k, v)
        seen[name] = O.process_frame(f, prm, tpl, nn_mode=1, want_clouds=True)
    a, b = seen["chain"]["result"], seen["cuboid"]["result"]
    assert (a.n_cropped, a.n_voxels, a.n_plane) == (b.n_cropped, b.n_voxels, b.n_plane)
    assert b.n_clusters == 1 and b.clusters[0].size == b.n_objects >= a.n_objects
    assert not seen["cuboid"]["labels"].any()


@pytest.mark.parametrize("flavour", sorted(FLAVOURS))
def test_oracle_against_real_pcl(O, flavour):
    fname, over = FLAVOURS[flavour]
    path = os.path.join(GOLDEN, fname)
    if not os.path.exists(path):
        pytest.skip("parity unpinned: tests/golden/%s does not exist - it takes a machine with PCL (tools/pcl_golden.cpp, flavour "
                    "'%s'); nothing in this repository may claim parity with real PCL until it does" % (fname, flavour))
    gold = json.load(open(path))
    assert gold.get("flavour", "chain") == flavour
    tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
    prm = capi.default_params()
    prm.rgb_offset = 12
    for k, v in over.items():
        setattr(prm, k, v)
    for e in gold["frames"]:
        f = synth.frame(e["index"])
        assert hashlib.sha256(f.tobytes()).hexdigest() == e["frame_sha256"], "the PCL run used other frames"
        r = O.process_frame(f, prm, tpl, nn_mode=1, want_clouds=True)
        res = r["result"]
        assert (res.n_cropped, res.n_voxels) == (e["n_cropped"], e["n_voxels"])
        # plane indices and cluster labels: exact (north_star); the digests cover the whole lists
        assert (res.n_plane, res.n_objects, res.n_clusters) == (e["n_plane"], e["n_objects"], e["n_clusters"])
        assert hashlib.sha256(r["plane_inliers"].astype(np.int32).tobytes()).hexdigest() == e["plane_inliers_sha256"]
        assert hashlib.sha256(r["labels"].astype(np.int32).tobytes()).hexdigest() == e["labels_sha256"]
        for k, c in enumerate(e["clusters"][:capi.CD_MAX_CLUSTERS_PER_FRAME]):
            a = res.clusters[k]
            assert a.size == c["size"]
            err = float(np.linalg.norm(np.array(a.pose) - np.array(c["pose"])))
            assert err < 1e-4, "frame %d cluster %d: pose differs from PCL's by %g (tolerance 1e-4)" % (e["index"], k, err)
