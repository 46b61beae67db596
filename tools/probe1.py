import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as g
g.smoke()
from perception_amd import capi, synth, templates
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
F = 16
fr = np.stack([synth.frame(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
ctx.set_template(0, tpl)
for rep in range(3):
    t0 = time.time(); res, _, _ = ctx.process_batch(fr, prm); dt = time.time() - t0
    t = ctx.timing()
    print('F=%d wall %.2f ms; stages' % (F, dt * 1e3), ['%.3f' % x for x in t.stage_ms], 'icp launches', t.icp_kernel_launches, 'icp ms %.3f' % t.icp_kernel_ms,
          'pairs', (t.icp_pair_tests_hi << 32) | (t.icp_pair_tests_lo & 0xffffffff), 'balg', t.algorithmic_bytes)
for rep in range(3):
    t0 = time.time(); res, _, _ = ctx.process_batch(fr[:1], prm); dt = time.time() - t0
    t = ctx.timing()
    print('F=1 wall %.2f ms; stages' % (dt * 1e3), ['%.3f' % x for x in t.stage_ms], 'icp launches', t.icp_kernel_launches, 'icp ms %.3f' % t.icp_kernel_ms)
print([ (r.n_clusters, [r.clusters[k].iterations for k in range(r.n_clusters)]) for r in res])
