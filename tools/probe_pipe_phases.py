"""Where the waves of k_icp_pipe / k_icp_pipe_big spend their time (needs a -DCD_TIMERS build:
   tools/build_variant.sh timers k_icp.hip -DCD_TIMERS; CUBOID_HIP_LIB=perception_amd/lib/variants/libtimers.so python tools/probe_pipe_phases.py [default|big|cfg5big] [F])
   phases: 0 epoch wait, 1 fetch + seeds, 2 grid walk, 3 solve / hand-over, 4 wave-per-query search, 5 moments + fold."""
import sys, os, time, ctypes as C, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from perception_amd import capi, synth, templates, pcd
which = sys.argv[1] if len(sys.argv) > 1 else "default"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "big":
    tpl = pcd.read_xyz(os.path.join(R, "tests", "golden", "template_cuboid_L200_W100_H75.pcd")).astype(np.float32)
elif which == "cfg5big":
    tpl = templates.template_xyz32(*synth.CONFIG5_DIMS[3])
else:
    tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
lib = capi.load_library()
prm = capi.default_params(); prm.rgb_offset = 12
fr = np.stack([synth.frame(i) for i in range(F)], 0)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
ctx.set_template(0, tpl)
d = torch.from_numpy(fr).cuda(); torch.cuda.synchronize()
res = (capi.CdFrameResult * F)()
out = (C.c_ulonglong * 16)()
ctx.process_batch_device(d.data_ptr(), 16, fr.shape[1], F, prm, results=res)
if hasattr(lib, "cd_debug_icp_stats"):
    lib.cd_debug_icp_stats(out, 1)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); ctx.process_batch_device(d.data_ptr(), 16, fr.shape[1], F, prm, results=res); ts.append(time.perf_counter() - t0)
tm = ctx.timing()
print("template %d points, F=%d: batch %.2f ms, icp stage %.2f ms (kernel %.2f ms, %d launches)" % (len(tpl), F, 1e3 * min(ts), tm.stage_ms[3], tm.icp_kernel_ms, tm.icp_kernel_launches))
if hasattr(lib, "cd_debug_icp_stats"):
    lib.cd_debug_icp_stats(out, 1)
    o = list(out)
    ph = o[8:14]
    tot = max(sum(ph), 1)
    names = ["epoch wait", "fetch+seeds", "grid walk", "solve/hand-over", "far search", "moments+fold"]
    print("CD_TIMERS (3 batches): " + "  ".join("%s %.1f%%" % (n, 100.0 * x / tot) for n, x in zip(names, ph)))
    if o[15]:
        print("workgroup busy time mean %.3f ms, max %.3f ms over %d workgroup launches (balance %.2f)" % (o[14] / max(o[7], 1) / 1e5, o[15] / 1e5, o[7], o[14] / max(o[7], 1) / max(o[15], 1)))
