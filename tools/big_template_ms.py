import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from perception_amd import capi, synth, templates, pcd
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
big = pcd.read_xyz(os.path.join(R, "tests", "golden", "template_cuboid_L200_W100_H75.pcd")).astype(np.float32)
print("template points", len(big))
prm = capi.default_params(); prm.rgb_offset = 12
for F in ([int(a) for a in sys.argv[1:]] or [1, 32]):
    fr = np.stack([synth.frame(i) for i in range(F)], 0)
    ctx = capi.Context(max_points=fr.shape[1], max_frames=F)
    ctx.set_template(0, big)
    d = torch.from_numpy(fr).cuda(); torch.cuda.synchronize()
    res = (capi.CdFrameResult * F)()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); ctx.process_batch_device(d.data_ptr(), 16, fr.shape[1], F, prm, results=res); ts.append(time.perf_counter() - t0)
    tm = ctx.timing()
    print("F=%d batch %.2f ms, icp stage %.2f ms, launches %d, iterations %s" % (F, 1e3 * min(ts[1:]), tm.stage_ms[3], tm.icp_kernel_launches, [res[0].clusters[k].iterations for k in range(res[0].n_clusters)]))
    ctx.close()
