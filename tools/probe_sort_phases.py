import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from perception_amd import capi, templates
F = 256
frames = bench.make_frames(0, F)
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params(); prm.rgb_offset = 12
d = torch.from_numpy(frames).cuda(); torch.cuda.synchronize()
ctx = capi.Context(max_points=frames.shape[1], max_frames=F); ctx.set_template(0, tpl)
res = (capi.CdFrameResult * F)()
lib = capi.load_library()
out = (C.c_ulonglong * 8)()
for rep in range(3):
    lib.cd_debug_sort(out, 1)
    ctx.process_batch_device(d.data_ptr(), 16, frames.shape[1], F, prm, results=res)
    lib.cd_debug_sort(out, 0)
    v = list(out); n = max(v[7], 1)
    print("scatter workgroups %d: us per workgroup: load+rank %.2f  prefix+chained scan %.2f  stage to LDS %.2f  write out %.2f" % (n, v[0]/n/100, v[1]/n/100, v[2]/n/100, v[3]/n/100))
