"""Seed-ball radius (in grid cells) of the queries the persistent ICP kernels hand to the wave-per-query search, by iteration class
   tools/build_variant.sh stats k_icp.hip -DCD_STATS; CUBOID_ICP_MODE=pipe CUBOID_HIP_LIB=perception_amd/lib/variants/libstats.so python tools/probe_far_radius.py [c3|c5] [F]"""
import sys, os, ctypes as C, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from perception_amd import capi, synth, templates
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
F = int(sys.argv[2]) if len(sys.argv) > 2 else (32 if which == "c3" else 2)
lib = capi.load_library()
prm = capi.default_params(); prm.rgb_offset = 12
if which == "c5":
    prm.template_slot = -1
    prm.crop_x_min, prm.crop_x_max = -synth.CONFIG5_CROP_X, synth.CONFIG5_CROP_X
    prm.crop_z_max = prm.crop2_z_max = 1.2
    fr = np.stack([synth.frame_config5(i) for i in range(F)], 0)
else:
    fr = np.stack([synth.frame(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
if which == "c5":
    for k, dims in enumerate(synth.CONFIG5_DIMS):
        ctx.set_template(k, templates.template_xyz32(*dims))
else:
    ctx.set_template(0, templates.template_xyz32(**templates.DEFAULT_TEMPLATE))
st = (C.c_ulonglong * 16)()
rh = (C.c_ulonglong * 48)()
lib.cd_debug_icp_stats(st, 1); lib.cd_debug_icp_rhist(rh, 1)
ctx.process_batch(fr, prm)
lib.cd_debug_icp_stats(st, 1); lib.cd_debug_icp_rhist(rh, 1)
q = max(st[0], 1)
print("%s, %d frames: %d queries, far %.3f" % (which, F, st[0], st[2] / q))
for c, nm in enumerate(("it < 3", "3 <= it < 16", "it >= 16")):
    h = np.array(rh[16 * c:16 * c + 16], dtype=np.float64)
    print("%-13s far queries %9d (%.3f of all queries)  radius in cells: %s" % (nm, h.sum(), h.sum() / q, " ".join("%d:%.2f" % (i, v / max(h.sum(), 1)) for i, v in enumerate(h) if v)))
