"""ICP stage time of the bench batch on an otherwise idle GPU (strictly serial), plus a bit-for-bit check of the records
against the k_icp_cluster driver (same arithmetic, simpler kernel).  usage: tools/icp_ms.py [frames] [reps]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
import bench
frames = bench.make_frames(0, F)
import torch
from perception_amd import capi, templates
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params()
prm.rgb_offset = 12
d = torch.from_numpy(frames).cuda()
torch.cuda.synchronize()
N = frames.shape[1]


def run(mode):
    if mode:
        os.environ["CUBOID_ICP_MODE"] = mode
    else:
        os.environ.pop("CUBOID_ICP_MODE", None)
    ctx = capi.Context(max_points=N, max_frames=F)
    ctx.set_template(0, tpl)
    res = (capi.CdFrameResult * F)()
    ts = []
    for _ in range(reps):
        ctx.process_batch_device(d.data_ptr(), 16, N, F, prm, results=res)
        t = ctx.timing()
        ts.append((t.icp_kernel_ms, list(t.stage_ms)))
    rec = capi.results_to_array(res).copy()
    ctx.close()
    return rec, ts


rec, ts = run(os.environ.get("CUBOID_ICP_MODE"))   # None: the library's own choice
icp = [t[0] for t in ts[1:]] or [ts[0][0]]
st = np.array([t[1] for t in ts[1:]] or [ts[0][1]])
print("icp_kernel_ms min %.3f avg %.3f | stages crop_voxel %.3f plane %.3f extract_cluster %.3f icp %.3f total %.3f" %
      ((min(icp), sum(icp) / len(icp)) + tuple(st.mean(0))))
if os.environ.get("ICP_MS_NOCHECK") != "1":
    ref, _ = run("cluster")
    same = np.array_equal(rec, ref)
    print("records identical to k_icp_cluster driver:", same, hashlib.sha256(rec.tobytes()).hexdigest()[:16])
    if not same:
        bad = [f for f in range(F) if not np.array_equal(rec[f], ref[f])]
        print("first differing frames:", bad[:10])
        sys.exit(1)
