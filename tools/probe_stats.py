import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from perception_amd import capi, synth, templates
lib = capi.load_library()
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
F = 16
fr = np.stack([synth.frame(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
ctx.set_template(0, tpl)
out = (C.c_ulonglong * 4)()
lib.cd_debug_icp_stats(out, 1)
res, _, _ = ctx.process_batch(fr, prm)
lib.cd_debug_icp_stats(out, 1)
t = ctx.timing()
tests, proc, lanes = out[0], out[1], out[2]
print('launches', t.icp_kernel_launches, 'icp ms', t.icp_kernel_ms)
print("runs offered", tests, "runs visited", proc, "queries", lanes, "visited per query", proc / max(lanes, 1), "frac", proc / max(tests, 1))
its = [r.clusters[k].iterations for r in res for k in range(r.n_clusters)]
print('iters', its)
