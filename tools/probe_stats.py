import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
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
passes = max(o[3] / 64, 1)
print('queries', o[0], 'near(grid)', o[3], 'far(wave-per-query)', o[2], 'runs visited per far query', o[1] / max(o[2], 1))
print('grid per 64-query pass: row-step iterations', o[4] / passes, '(active lanes', o[5] / max(o[4], 1), ') point-test iterations', o[6] / passes, '(active lanes', o[7] / max(o[6], 1), ')')
print('grid per near query: row steps', o[5] / max(o[3], 1), 'points tested', o[7] / max(o[3], 1))
ph = o[8:14]
tot = max(sum(ph), 1)
print('wave-0 cycles by phase: fetch %.1f%% grid %.1f%% far %.1f%% moments %.1f%% block_sum %.1f%% solve+barrier %.1f%%  (total %.3g cycles)' % tuple([100.0 * x / tot for x in ph] + [tot]))
