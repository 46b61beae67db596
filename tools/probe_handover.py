"""Timeline of the hand-overs of running clusters in k_icp_pipe (needs a -DCD_DONDBG build:
   tools/build_variant.sh dondbg k_icp.hip -DCD_DONDBG; CUBOID_HIP_LIB=perception_amd/lib/variants/libdondbg.so python tools/probe_handover.py [F])"""
import sys, os, ctypes as C, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
from perception_amd import capi, templates
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
lib = capi.load_library()
prm = capi.default_params(); prm.rgb_offset = 12
fr = bench.make_frames(0, F)
ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
ctx.set_template(0, tpl)
d = torch.from_numpy(fr).cuda(); torch.cuda.synchronize()
res = (capi.CdFrameResult * F)()
out = (C.c_ulonglong * 64)()
ctx.process_batch_device(d.data_ptr(), 16, fr.shape[1], F, prm, results=res)
lib.cd_debug_don(out, 1)
ctx.process_batch_device(d.data_ptr(), 16, fr.shape[1], F, prm, results=res)
lib.cd_debug_don(out, 1)
o = list(out)
print("icp kernel %.3f ms; last workgroup ends at %.3f ms; workgroups waited %.1f ms in all" % (ctx.timing().icp_kernel_ms, o[1] / 1e5, o[2] / 1e5))
print("quarter-ms bin :", " ".join("%4d" % b for b in range(16)))
print("waits begun    :", " ".join("%4d" % v for v in o[8:24]))
print("published      :", " ".join("%4d" % v for v in o[24:40]))
print("taken          :", " ".join("%4d" % v for v in o[40:56]))
