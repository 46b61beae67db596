"""Smallest end-to-end check of one ICP mode against the oracle (run with CUBOID_ICP_MODE=...)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from perception_amd import capi, synth, templates
from oracle import oracle_py as O
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
fr = np.stack([synth.frame(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
ctx.set_template(0, tpl)
res, _, _ = ctx.process_batch(fr, prm)
bad = 0
for f in range(F):
    o = O.process_frame(fr[f], prm, tpl)["result"]
    for k in range(min(o.n_clusters, 8)):
        a, b = res[f].clusters[k], o.clusters[k]
        same = list(a.T) == list(b.T) and a.iterations == b.iterations and a.fitness == b.fitness
        bad += not same
        print(f, k, a.iterations, b.iterations, a.fitness, b.fitness, 'OK' if same else 'MISMATCH')
print('mismatches', bad, 'icp ms', ctx.timing().icp_kernel_ms)
sys.exit(1 if bad else 0)
