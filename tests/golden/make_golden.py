#!/usr/bin/env python3
"""Generates tests/golden/* .  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

What it writes (all DATA, no reference source text):
  reference_templates.json  SHA-256 + point count of the four committed template PCDs, and
                            of the bytes the reference's own make_cuboid.py (executed here,
                            in a temp dir, as a subprocess) writes for the two templates it
                            still reproduces.
  template_cuboid_L200_W75_H100_3faces.pcd, template_cuboid_L200_W100_H75_3faces.pcd
                            the two small reference templates (data files, 46-50 KB).
  template_cuboid_L200_W100_H75.pcd
                            the 21 400-point six-face template of an earlier generator version (data file, 578 KB):
                            a real ICP target that does not fit the LDS image.
  {marker,screwdriver,eraser,clamp}_ascii.pcd and *_ascii_tf.pcd
                            real D435 object clusters held by the reference and their transformed copies (data files).
  transforms.json           the (translation, quaternion) values of transforms.txt.
  frames_golden.json        the ORACLE's outputs on synthetic frames 0..3 (so the GPU box,
                            which has no /root/reference, can check both the oracle and the
                            HIP path against committed numbers).
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def sha(b):
    return hashlib.sha256(b).hexdigest()


def reference_templates():
    tdir = os.path.join(REF, "cuboid_detection", "templates")
    out = {"committed": {}, "make_cuboid_py": {}}
    for fn in sorted(os.listdir(tdir)):
        if fn.endswith(".pcd"):
            b = open(os.path.join(tdir, fn), "rb").read()
            npts = int(re.search(rb"POINTS (\d+)", b).group(1))
            out["committed"][fn] = {"sha256": sha(b), "points": npts, "bytes": len(b)}
    # run the reference's generator (Python reference, importable/executable here)
    with tempfile.TemporaryDirectory() as tmp:
        for args in (["-L", "0.2", "-W", "0.1", "-H", "0.03", "-d", "0.002"],
                     ["-L", "0.2", "-W", "0.075", "-H", "0.1", "-d", "0.005"]):
            subprocess.run([sys.executable, os.path.join(tdir, "make_cuboid.py")] + args, cwd=tmp, check=True,
                           stdout=subprocess.DEVNULL)
        for fn in sorted(os.listdir(tmp)):
            b = open(os.path.join(tmp, fn), "rb").read()
            out["make_cuboid_py"][fn] = {"sha256": sha(b), "bytes": len(b)}
    for fn in ("template_cuboid_L200_W75_H100_3faces.pcd", "template_cuboid_L200_W100_H75_3faces.pcd",
               "template_cuboid_L200_W100_H75.pcd"):
        shutil.copyfile(os.path.join(tdir, fn), os.path.join(HERE, fn))
    return out


def object_fixtures():
    odir = os.path.join(REF, "object_detection", "templates")
    for fn in ("marker_ascii.pcd", "marker_ascii_tf.pcd", "screwdriver_ascii.pcd", "screwdriver_ascii_tf.pcd",
               "eraser_ascii.pcd", "eraser_ascii_tf.pcd", "clamp_ascii.pcd", "clamp_ascii_tf.pcd"):
        shutil.copyfile(os.path.join(odir, fn), os.path.join(HERE, fn))
    txt = open(os.path.join(odir, "transforms.txt")).read()
    out = {}
    for name, block in re.findall(r"#+ (\w+) #+\n(.*?)(?=\n#+ \w+ #+|\Z)", txt, flags=re.S):
        vals = {}
        for sect in ("translation", "rotation"):
            m = re.search(sect + r":\s*\n((?:\s+[xyzw]: [-\d.eE]+\n?)+)", block)
            vals[sect] = {k: float(v) for k, v in re.findall(r"([xyzw]): ([-\d.eE]+)", m.group(1))}
        out[name] = vals
    return out


def frames_golden():
    import numpy as np
    from oracle import oracle_py as O
    from perception_amd import capi, synth, templates
    tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
    prm = capi.default_params()
    prm.rgb_offset = 12
    out = {"params": "capi.default_params() with rgb_offset=12", "frames": []}
    for i in range(4):
        f = synth.frame(i)
        r = O.process_frame(f, prm, tpl, nn_mode=1, want_clouds=True)
        res = r["result"]
        e = {
            "index": i, "frame_sha256": sha(f.tobytes()),
            "n_cropped": res.n_cropped, "n_voxels": res.n_voxels, "n_plane": res.n_plane,
            "n_objects": res.n_objects, "n_clusters": res.n_clusters,
            "ransac_iterations": res.ransac_iterations,
            "plane_hex": [float(x).hex() for x in res.plane],
            "voxels_sha256": sha(r["voxels"].tobytes()),
            "plane_inliers_sha256": sha(r["plane_inliers"].tobytes()),
            "labels_sha256": sha(r["labels"].tobytes()),
            "clusters": [],
        }
        for k in range(min(res.n_clusters, capi.CD_MAX_CLUSTERS_PER_FRAME)):
            c = res.clusters[k]
            e["clusters"].append({"size": c.size, "iterations": c.iterations, "converged": c.converged,
                                  "accepted": c.accepted, "fitness_hex": float(c.fitness).hex(),
                                  "T_hex": [float(x).hex() for x in c.T],
                                  "pose": [float(x) for x in c.pose]})
        out["frames"].append(e)
    return out


if __name__ == "__main__":
    json.dump(reference_templates(), open(os.path.join(HERE, "reference_templates.json"), "w"), indent=1, sort_keys=True)
    json.dump(object_fixtures(), open(os.path.join(HERE, "transforms.json"), "w"), indent=1, sort_keys=True)
    json.dump(frames_golden(), open(os.path.join(HERE, "frames_golden.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)
