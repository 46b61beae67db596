"""Phase times of k_cluster_lds (needs a -DCD_CLDBG build of k_cluster.hip:
tools/build_variant.sh cl_dbg k_cluster.hip -DCD_CLDBG, then CUBOID_HIP_LIB=perception_amd/lib/variants/libcl_dbg.so)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from perception_amd import capi, templates
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = bench.make_frames(0, F)
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
prm.rgb_offset = 12
d = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
ctx = capi.Context(max_points=frames.shape[1], max_frames=F)
ctx.set_template(0, tpl)
lib = capi.load_library()
res = (capi.CdFrameResult * F)()
out = (C.c_ulonglong * 16)()
for rep in range(3):
    lib.cd_debug_cluster(out, 1)
    ctx.process_batch_device(d.data_ptr(), 16, frames.shape[1], F, prm, results=res)
    lib.cd_debug_cluster(out, 0)
    v = np.array(list(out)).reshape(8, 2)   # per phase: sum over the workgroups, max over the workgroups (100 MHz ticks)
    print("k_cluster_lds us per workgroup (mean / max): table + cell sort %.1f / %.1f   hook %.1f / %.1f   components + store %.1f / %.1f"
          % (v[0, 0] / F / 100, v[0, 1] / 100, v[1, 0] / F / 100, v[1, 1] / 100, v[2, 0] / F / 100, v[2, 1] / 100))
print("n_objects per frame: min %d mean %.0f max %d" % (min(r.n_objects for r in res), np.mean([r.n_objects for r in res]), max(r.n_objects for r in res)))
