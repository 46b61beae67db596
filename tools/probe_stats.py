"""Counters / phase timers of the ICP kernels (needs a -DCD_STATS or -DCD_TIMERS build of perception_amd/csrc)."""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perception_amd import capi, synth, templates
lib = capi.load_library()
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
fr = np.stack([synth.frame(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
ctx.set_template(0, tpl)
out = (C.c_ulonglong * 16)()
lib.cd_debug_icp_stats(out, 1)
res, _, _ = ctx.process_batch(fr, prm)
lib.cd_debug_icp_stats(out, 1)
o = list(out)
q = max(o[0], 1)
passes = max(q / 64.0, 1)
print('CD_STATS: queries %d  near (grid walk) %.3f  far (wave-per-query) %.3f, runs visited per far query %.2f' % (o[0], o[3] / q, o[2] / q, o[1] / max(o[2], 1)))
print('CD_STATS: grid walk per 64-query pass: row-step iterations %.1f (active lanes %.1f), point iterations %.1f (active lanes %.1f)' % (o[4] / passes, o[5] / max(o[4], 1), o[6] / passes, o[7] / max(o[6], 1)))
print('CD_STATS: per near query: row steps %.2f, points tested %.2f' % (o[5] / max(o[3], 1), o[7] / max(o[3], 1)))
ph = o[8:14]
tot = max(sum(ph), 1)
print('CD_TIMERS: wave cycles by phase ' + ' '.join('%.1f%%' % (100.0 * x / tot) for x in ph), 'total %.3g' % tot)
if o[15]:
    print('CD_TIMERS: workgroup busy time mean %.3f ms, max %.3f ms over %d workgroups (balance %.2f)' % (o[14] / max(o[7], 1) / 1e5, o[15] / 1e5, o[7], o[14] / max(o[7], 1) / max(o[15], 1)))
