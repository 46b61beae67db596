"""Phase cycles of k_cluster_lds (needs make -C perception_amd/csrc FLAGS_EXTRA=-DCD_TIMERS)."""
import sys, ctypes as C, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perception_amd import capi, synth, templates
lib = capi.load_library()
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
F = 64
fr = np.stack([synth.frame(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
ctx.set_template(0, tpl)
res, _, _ = ctx.process_batch(fr, capi.default_params())
out = (C.c_ulonglong * 8)()
lib.cd_debug_cluster_stats(out)
o = list(out); t = max(sum(o[:3]), 1)
print('k_cluster_lds thread-0 cycles: build lists %.1f%%  neighbour/union loop %.1f%%  flatten+sizes+store %.1f%%  (total %.3g over %d frames)' % (100*o[0]/t, 100*o[1]/t, 100*o[2]/t, t, F))
print('n_objects per frame: min %d mean %.0f max %d' % (min(r.n_objects for r in res), np.mean([r.n_objects for r in res]), max(r.n_objects for r in res)))
print('per point: candidates walked %.1f  pairs within radius %.1f  CAS attempts %.2f' % (o[4]/max(o[7],1), o[5]/max(o[7],1), o[6]/max(o[7],1)))
