import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perception_amd import capi, synth, templates
from oracle import oracle_py as O
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
ctx = capi.Context(max_points=307200, max_frames=1)
for i in (0, 1, 5, 9, 13):
    sc = synth.scene_for(i)
    r = O.process_frame(synth.frame(i), prm, tpl, want_clouds=True)
    obj = r['objects']
    for rep in range(2):
        t0 = time.perf_counter(); lab, sizes, k = ctx.cluster(obj, prm); dt = time.perf_counter() - t0
    print('frame', i, 'boxes', len(sc['boxes']), 'n_o', len(obj), 'K', k, 'call ms %.3f' % (dt * 1e3))
# dense plane patch near the cap
g = np.stack(np.meshgrid(np.arange(90), np.arange(90)), -1).reshape(-1, 2) * 0.005
pl = np.zeros((len(g), 3), np.float32); pl[:, :2] = g; pl[:, 2] = 0.5
t0 = time.perf_counter(); lab, sizes, k = ctx.cluster(pl, prm); dt = time.perf_counter() - t0
print('dense plane n', len(pl), 'K', k, 'call ms %.3f' % (dt * 1e3))
