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


@pytest.mark.skipif(not os.path.exists(PCL_GOLDEN),
                    reason="parity unpinned: tests/golden/pcl_frames_golden.json does not exist - it takes a machine with PCL "
                           "(tools/pcl_golden.cpp); nothing in this repository may claim parity with real PCL until it does")
def test_oracle_against_real_pcl(O):
    gold = json.load(open(PCL_GOLDEN))
    tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
    prm = capi.default_params()
    prm.rgb_offset = 12
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
