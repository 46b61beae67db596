"""Single-frame latency breakdown (BASELINE config 2)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from perception_amd import capi, synth, templates
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params(); prm.rgb_offset = 12
fr = np.stack([synth.frame(i) for i in range(4)], 0)
N = fr.shape[1]
ctx = capi.Context(max_points=N, max_frames=1)
ctx.set_template(0, tpl)
d = torch.from_numpy(fr).cuda(); torch.cuda.synchronize()
one = (capi.CdFrameResult * 1)()
for f in range(4):
    lat = []
    for _ in range(6):
        a = time.perf_counter()
        ctx.process_batch_device(d[f].data_ptr(), 16, N, 1, prm, results=one)
        lat.append((time.perf_counter() - a) * 1e3)
    t = ctx.timing()
    print('frame', f, 'clusters', one[0].n_clusters, 'iters', [one[0].clusters[k].iterations for k in range(one[0].n_clusters)],
          'latency ms min %.3f' % min(lat), 'stages', ['%.3f' % x for x in t.stage_ms], 'icp kernels %.3f ms in %d launches' % (t.icp_kernel_ms, t.icp_kernel_launches))
