"""Executed work of the dominant kernel (k_icp_pipe) on the bench batch, counted by a -DCD_STATS build, written to
profiles/r03_icp_work.json (bench.py's roofline.valu block reads it).
   tools/build_variant.sh stats k_icp.hip -DCD_STATS
   CUBOID_HIP_LIB=perception_amd/lib/variants/libstats.so python tools/probe_icp_work.py [out.json]
A "pair test" is one evaluation of the canonical squared distance (3 subtractions, 3 multiplications, 2 additions = 8 flops)
between a query and a template point by one lane - executed work, including the lanes of a wave that test a point they do not
need (the loops are branch-free) - plus the box tests of the wave-per-query search (a box lower bound = 6 sub, 6 max, 3 mul,
2 add = 17 flops by each of the 64 lanes, two boxes per far query when both k-d halves are in reach)."""
import sys, os, json, ctypes as C, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from perception_amd import capi, templates
F = 256
frames = bench.make_frames(0, F)
lib = capi.load_library()
assert hasattr(lib, "cd_debug_icp_stats"), "needs a -DCD_STATS build (CUBOID_HIP_LIB)"
tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE)
prm = capi.default_params(); prm.rgb_offset = 12
ctx = capi.Context(max_points=frames.shape[1], max_frames=F)
ctx.set_template(0, tpl)
out = (C.c_ulonglong * 16)()
ctx.process_batch(frames, prm)
lib.cd_debug_icp_stats(out, 1)
res, _, _ = ctx.process_batch(frames, prm)
lib.cd_debug_icp_stats(out, 1)
o = [int(v) for v in out]
ncl = sum(min(r.n_clusters, 8) for r in res)
qi = sum(r.clusters[k].size * (r.clusters[k].iterations + 1) for r in res for k in range(min(r.n_clusters, 8)))
grid_tests = o[6] * 64 * 2          # every trip of the point loop: 64 lanes x 2 points
patch_tests = o[1] * 64             # every patch visit: 64 lanes x 1 point
seed_tests = o[8]
box_tests = o[2] * 64 * 2           # upper bound: two boxes per lane per far query
pair = grid_tests + patch_tests + seed_tests
d = {"_how": "tools/probe_icp_work.py on a -DCD_STATS build, one 256-frame bench batch (serial, one context)",
     "clusters": ncl, "query_iterations": qi, "queries_counted": o[0], "near_queries": o[3], "far_queries": o[2],
     "row_step_wave_iterations": o[4], "point_loop_wave_trips": o[6], "patches_visited": o[1],
     "executed_pair_tests": {"grid_walk": grid_tests, "patch_search": patch_tests, "seeds": seed_tests, "total": pair},
     "executed_box_tests_upper_bound": box_tests,
     "flops_per_pair_test": 8, "flops_per_box_test": 17,
     "executed_flops": pair * 8 + box_tests * 17,
     "bruteforce_equivalent_pair_tests": qi * len(tpl)}
p = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "gpurun_out", "r03_icp_work.json")
json.dump(d, open(p, "w"), indent=1)
print(json.dumps(d))
