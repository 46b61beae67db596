"""Stress of the hand-over of running clusters (k_icp_pipe / k_icp_pipe_big): many launch shapes, every record compared byte for
byte with a launch without hand-overs.  usage: tools/stress_handover.py [frames] [repeats]"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["CUBOID_ICP_MODE"] = "pipe"
from perception_amd import capi, synth, templates, pcd
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
frames = np.stack([synth.frame(i) for i in range(100, 100 + nf)], 0)
prm = capi.default_params()
tpls = {"lds": templates.template_xyz32(**templates.DEFAULT_TEMPLATE),
        "big": pcd.read_xyz(os.path.join(R, "tests", "golden", "template_cuboid_L200_W100_H75.pcd")).astype(np.float32)}
bad = total = moved = 0
for name, tpl in tpls.items():
    def run(wg, slots, donate):
        os.environ.update(CUBOID_ICP_MAX_WG=str(wg), CUBOID_ICP_SLOTS=str(slots), CUBOID_ICP_DONATE=str(donate))
        ctx = capi.Context(max_points=frames.shape[1], max_frames=nf)
        ctx.set_template(0, tpl)
        out, h = [], 0
        for _ in range(reps if donate else 1):
            res, _, _ = ctx.process_batch(frames, prm)
            out.append(capi.results_to_array(res).copy()); h += ctx.timing().icp_handovers
        ctx.close()
        return out, h
    ref, _ = run(4, 2, 0)
    for wg, slots in itertools.product((2, 3, 5, 8, 13, 21, 34), (2, 3, 4)):
        out, h = run(wg, slots, 1)
        moved += h
        for o in out:
            total += 1
            if not np.array_equal(o, ref[0]):
                bad += 1
                print("MISMATCH", name, "workgroups", wg, "slots", slots)
    print(name, "done: hand-overs so far", moved)
print("launches compared %d, mismatching %d, clusters handed over %d" % (total, bad, moved))
sys.exit(1 if bad else 0)
