"""CD_STATS counters of the persistent ICP kernels on BASELINE config 5 frames (every cluster x the LDS-resident templates, or all five)
   tools/build_variant.sh stats k_icp.hip -DCD_STATS; CUBOID_ICP_MODE=pipe CUBOID_HIP_LIB=perception_amd/lib/variants/libstats.so python tools/probe_stats_c5.py [lds|all] [F]"""
import sys, os, ctypes as C, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from perception_amd import capi, synth, templates
which = sys.argv[1] if len(sys.argv) > 1 else "lds"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = capi.load_library()
prm = capi.default_params(); prm.rgb_offset = 12
prm.template_slot = -1
prm.crop_x_min, prm.crop_x_max = -synth.CONFIG5_CROP_X, synth.CONFIG5_CROP_X
prm.crop_z_max = prm.crop2_z_max = 1.2
fr = np.stack([synth.frame_config5(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
for k, dims in enumerate(synth.CONFIG5_DIMS):
    if which == "all" or k != 3:
        ctx.set_template(k, templates.template_xyz32(*dims))
out = (C.c_ulonglong * 16)()
lib.cd_debug_icp_stats(out, 1)
res, _, _ = ctx.process_batch(fr, prm)
lib.cd_debug_icp_stats(out, 1)
o = list(out)
q = max(o[0], 1)
passes = max(q / 64.0, 1)
tm = ctx.timing()
print("config 5, %d frames, templates %s: icp stage %.2f ms" % (F, which, tm.stage_ms[3]))
print('queries %d  near (grid walk) %.3f  far (wave-per-query) %.3f, patches visited per far query %.2f' % (o[0], o[3] / q, o[2] / q, o[1] / max(o[2], 1)))
print('grid walk per 64-query pass: row-step iterations %.1f (active lanes %.1f), point iterations %.1f (active lanes %.1f)' % (o[4] / passes, o[5] / max(o[4], 1), o[6] / passes, o[7] / max(o[6], 1)))
