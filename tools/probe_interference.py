"""What does a resident background kernel take away from the FRONT END (crop .. clusters; the ICP is cut to one iteration)?
Background: 522 workgroups x 256 threads for ~60 ms on a second stream - parked (s_sleep), a vector FMA chain, or a chain of L2
loads/stores - at three register footprints.  Prints the front end's stage times beside each.  tools/bg_load.hip"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
frames = bench.make_frames(0, 256)
import torch
from perception_amd import capi, templates
bg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libbg_load.so"))
bg.bg_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
prm.rgb_offset = 12
prm.icp_max_iterations = 1
d = torch.from_numpy(frames).cuda()
buf = torch.zeros(1 << 22, dtype=torch.float32, device="cuda")
sink = torch.zeros(4, dtype=torch.float32, device="cuda")
N = frames.shape[1]
ctx = capi.Context(max_points=N, max_frames=256)
ctx.set_template(0, tpl)
res = (capi.CdFrameResult * 256)()
side = torch.cuda.Stream()
for _ in range(3):
    ctx.process_batch_device(d.data_ptr(), 16, N, 256, prm, results=res)


def front(n=8):
    ts = []
    for _ in range(n):
        ctx.process_batch_device(d.data_ptr(), 16, N, 256, prm, results=res)
        ts.append(list(ctx.timing().stage_ms))
    return np.median(np.array(ts), 0)


print("alone: crop_voxel %.2f plane %.2f extract_cluster %.2f icp(1 it) %.2f total %.2f ms" % tuple(front()))
for n_wg in (522, 1044):
    for vg in (1, 40, 100):
        for kind, name in ((0, "parked"), (1, "fma chain"), (2, "L2 load/store chain")):
            torch.cuda.synchronize()
            bg.bg_launch(ctypes.c_void_p(side.cuda_stream), kind, n_wg, vg, 60.0, ctypes.c_void_p(buf.data_ptr()), buf.numel(), ctypes.c_void_p(sink.data_ptr()))
            time.sleep(0.002)
            t = front(6)
            torch.cuda.synchronize()
            print("background %4d workgroups x 4 waves, ~%3d registers, %-20s: crop_voxel %.2f plane %.2f extract_cluster %.2f icp(1 it) %.2f total %.2f ms" % ((n_wg, vg, name) + tuple(t)))
ctx.close()
