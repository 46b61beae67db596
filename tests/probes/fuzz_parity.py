import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import torch
from oracle import oracle_py as O
from perception_amd import capi, synth, templates
import test_gpu_fuzz as T
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.RandomState(seed)
ctx = capi.Context(max_points=synth.WIDTH * synth.HEIGHT, max_frames=2)
ctx.set_template(0, tpl)
bad = 0
for trial in range(n):
    prm = capi.default_params(); prm.rgb_offset = 12
    prm.leaf_size = float(rng.choice([0.002, 0.003, 0.004, 0.005, 0.006, 0.0075, 0.01, 0.015, 0.02, 0.03]))
    a, b = sorted(rng.uniform(-0.4, 0.4, 2).tolist())
    if b - a < 0.1: a, b = -0.25, 0.2
    prm.crop_x_min, prm.crop_x_max = a, b
    prm.crop_z_min = float(rng.choice([0.0, 0.05, 0.2, 0.4])); prm.crop_z_max = float(rng.uniform(0.55, 1.5))
    prm.plane_distance_threshold = float(rng.choice([0.003, 0.005, 0.01, 0.015, 0.03, 0.05]))
    prm.cluster_tolerance = float(rng.choice([0.006, 0.008, 0.012, 0.02, 0.035, 0.05, 0.08]))
    prm.cluster_min_size = int(rng.choice([1, 5, 20, 100, 200])); prm.cluster_max_size = int(rng.choice([100, 300, 1500, 25000]))
    frame = synth.frame(int(rng.randint(0, 256)))
    if trial % 7 == 3:   # sprinkle NaNs and far points
        frame = frame.copy(); idx = rng.randint(0, len(frame), 500); frame[idx, rng.randint(0, 3, 500)] = np.nan
    try:
        rg, pi, lb = ctx.process_frame(frame, prm, want_indices=True)
    except capi.CuboidError as e:
        o = O.process_frame(frame, prm, tpl)
        if o["status"] != e.status: bad += 1; print("trial", trial, "status differs", e.status, o["status"])
        continue
    o = O.process_frame(frame, prm, tpl, want_clouds=True)
    try:
        T._same_record(rg, o["result"])
        assert np.array_equal(pi[:rg.n_plane], o["plane_inliers"]) and np.array_equal(lb[:rg.n_objects], o["labels"])
    except AssertionError as e:
        bad += 1
        print("trial", trial, "MISMATCH", str(e)[:200], prm.leaf_size, prm.cluster_tolerance)
print("seed", seed, "trials", n, "mismatches", bad)
