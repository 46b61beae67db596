"""Every cluster against every template (template_slot = -1) with templates that all fit LDS: the grouped k_icp_pipe launch
(auto mode) against the sliced driver, same records.  usage: tools/multi_template_ms.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from perception_amd import capi, templates
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = bench.make_frames(0, F)
tpls = [templates.template_xyz32(**templates.DEFAULT_TEMPLATE), templates.template_xyz32(0.1, 0.05, 0.05, 0.002),
        templates.template_xyz32(0.05, 0.1, 0.05, 0.002), templates.template_xyz32(0.1, 0.1, 0.1, 0.002)]
prm = capi.default_params()
prm.rgb_offset = 12
prm.template_slot = -1
d = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
out = {}
for mode in ("auto", "sliced"):
    if mode == "auto":
        os.environ.pop("CUBOID_ICP_MODE", None)
    else:
        os.environ["CUBOID_ICP_MODE"] = mode
    ctx = capi.Context(max_points=frames.shape[1], max_frames=F)
    for s, t in enumerate(tpls):
        ctx.set_template(s, t)
    res = (capi.CdFrameResult * F)()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.process_batch_device(d.data_ptr(), 16, frames.shape[1], F, prm, results=res)
        ts.append(time.perf_counter() - t0)
        tm = ctx.timing()
    out[mode] = capi.results_to_array(res).copy()
    print("%-6s batch %.2f ms  icp stage %.2f ms  icp launches %d  templates %s" % (mode, 1e3 * min(ts[1:]), tm.stage_ms[3], tm.icp_kernel_launches, [len(t) for t in tpls]))
    ctx.close()
print("records identical:", np.array_equal(out["auto"], out["sliced"]))
sys.exit(0 if np.array_equal(out["auto"], out["sliced"]) else 1)
