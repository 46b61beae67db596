"""Where k_icp_lat's time goes (a -DCD_LAT_TIMERS build: tools/build_variant.sh lattimers k_icp_lat.hip -DCD_LAT_TIMERS):
thread 0's cycles per phase, summed over the workgroups of one 256-frame batch, per launch shape.
usage: CUBOID_HIP_LIB=perception_amd/lib/variants/liblattimers.so python tools/probe_lat_phases.py [shape ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
frames = bench.make_frames(0, 256)
import torch
from perception_amd import capi, templates
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
prm.rgb_offset = 12
d = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
N = frames.shape[1]
lib = capi.load_library()
names = ["work", "wait1", "solve", "wait2", "rounds", "stage"]
for shape in sys.argv[1:] or ["1,4"]:
    os.environ["CUBOID_LAT_SHAPE"] = shape
    ctx = capi.Context(max_points=N, max_frames=256)
    ctx.set_template(0, tpl)
    res = (capi.CdFrameResult * 256)()
    ctx.process_batch_device(d.data_ptr(), 16, N, 256, prm, results=res)
    buf = (ctypes.c_ulonglong * 8)()
    lib.cd_debug_lat_stats(buf, 1)
    ctx.process_batch_device(d.data_ptr(), 16, N, 256, prm, results=res)
    t = ctx.timing()
    lib.cd_debug_lat_stats(buf, 1)
    v = list(buf)
    rounds = max(v[4], 1)
    tot = sum(v[k] for k in (0, 1, 2, 3, 5))
    print("shape %-6s kernel %.3f ms, %d rounds (thread 0 of every workgroup): cycles per round %s | share %s" % (
        shape, t.icp_kernel_ms, rounds, {names[k]: round(v[k] / rounds) for k in (0, 1, 2, 3, 5)},
        {names[k]: round(v[k] / tot, 3) for k in (0, 1, 2, 3, 5)}))
    ctx.close()
