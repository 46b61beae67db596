"""Per-iteration cost of k_icp_pipe (needs a -DCD_ITSTATS build: make -C perception_amd/csrc FLAGS_EXTRA=-DCD_ITSTATS)."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = bench.make_frames(0, F)
from perception_amd import capi, templates
lib = capi.load_library()
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
ctx = capi.Context(max_points=frames.shape[1], max_frames=F)
ctx.set_template(0, tpl)
out = (C.c_ulonglong * 128)()
res, _, _ = ctx.process_batch(frames, prm)
lib.cd_debug_icp_it(out, 1)
res, _, _ = ctx.process_batch(frames, prm)
lib.cd_debug_icp_it(out, 1)
o = np.array(list(out), dtype=np.float64).reshape(32, 4)
tot = o[:, 0].sum()
print("it  passes   cyc/pass  share  far/pass  patches/far")
for i in range(32):
    if o[i, 1] > 0:
        print("%2d %8d %9.0f %6.3f %8.2f %8.2f" % (i, o[i, 1], o[i, 0] / o[i, 1], o[i, 0] / tot, o[i, 2] / o[i, 1], o[i, 3] / max(o[i, 2], 1)))
