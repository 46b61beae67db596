"""How far apart do the workgroups of the persistent ICP launch finish, and what would a better pairing of clusters buy?
Runs the bench batch, reads every cluster's size and iteration count, takes cost = size x (iterations + 1) passes-worth and
simulates W workgroups of S slots: (a) slots filled in queue order (workgroup i: items i, W + i, ...), (b) snake order
(i, 2W - 1 - i, 2W + i, ...), (c) the mean (perfect balance).  usage: tools/probe_balance.py [frames] [workgroups] [slots]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import torch
from perception_amd import capi, templates
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2
frames = bench.make_frames(0, F)
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params(); prm.rgb_offset = 12
d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
ctx = capi.Context(max_points=frames.shape[1], max_frames=F)
ctx.set_template(0, tpl)
res = (capi.CdFrameResult * F)()
ctx.process_batch_device(d.data_ptr(), 16, frames.shape[1], F, prm, results=res)
cl = []
for f in range(F):
    for c in ctx.cluster_results(f):
        if c.size >= 3:
            cl.append((c.size, c.iterations, float(np.linalg.norm(np.array(c.pose).reshape(4, 4)[:3, 3])), float(np.degrees(np.arccos(np.clip((np.trace(np.array(c.pose).reshape(4, 4)[:3, :3]) - 1) / 2, -1, 1))))))
cl.sort(key=lambda t: -t[0])
import json
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out'), exist_ok=True)
json.dump([[c[0], c[1]] for c in cl], open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'cluster_costs.json'), 'w'))
n = np.array([c[0] for c in cl], float); it = np.array([c[1] for c in cl], float)
cost = np.ceil(n / 64) * (it + 1)
M = len(cl)
tr = np.array([c[2] for c in cl]); ang = np.array([c[3] for c in cl])
print("corr(iterations, |translation of the final pose|) %.2f, corr(iterations, rotation angle) %.2f" % (np.corrcoef(it, tr)[0, 1], np.corrcoef(it, ang)[0, 1]))
for lo, hi in ((0, 20), (20, 40), (40, 60), (60, 80), (80, 99), (99, 1000)):
    m = (it >= lo) & (it < hi)
    if m.any(): print("  iterations [%d, %d): %d clusters, |t| mean %.3f, angle mean %.1f deg, size mean %.0f" % (lo, hi, m.sum(), tr[m].mean(), ang[m].mean(), n[m].mean()))
print("clusters %d, size mean %.0f (min %d max %d), iterations mean %.1f (min %d max %d), corr(size, iterations) %.2f" % (M, n.mean(), n.min(), n.max(), it.mean(), it.min(), it.max(), np.corrcoef(n, it)[0, 1]))
def sim(assign):
    load = np.zeros(W)
    rest = []
    taken = np.zeros(M, bool)
    for s in range(S):
        for i in range(W):
            k = assign(s, i)
            if k < M and not taken[k]:
                load[i] += cost[k]; taken[k] = True
    for k in np.nonzero(~taken)[0]:      # leftovers: to the least loaded (a refill goes to whoever finishes first)
        load[np.argmin(load)] += cost[k]
    return load
a = sim(lambda s, i: s * W + i)
b = sim(lambda s, i: s * W + i if s % 2 == 0 else (s + 1) * W - 1 - i)
print("queue order : max %.0f mean %.0f  (max / mean %.3f)" % (a.max(), a.mean(), a.max() / a.mean()))
print("snake order : max %.0f mean %.0f  (max / mean %.3f)" % (b.max(), b.mean(), b.max() / b.mean()))
# oracle pairing on the true cost (what knowing the iteration counts would give): LPT on cost
order = np.argsort(-cost); load = np.zeros(W); cnt = np.zeros(W, int)
for k in order:
    cand = np.where(cnt < S + 1, load, np.inf); j = np.argmin(cand); load[j] += cost[k]; cnt[j] += 1
print("LPT on true cost: max %.0f  (max / mean %.3f)" % (load.max(), load.max() / load.mean()))
