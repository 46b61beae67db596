"""Merge two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/r01_pmc_traffic.json.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Counters are in KB; on gfx950 FETCH_SIZE reads half of a wide coalesced read (MI355X_MICROARCH.md,
calibrated on k_crop_count which streams exactly frames*points*16 B), so hbm = (2*FETCH + WRITE)*1024.
"""
import collections
import csv
import glob
import json
import re
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"<.*>$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].split("(")[0].replace("cd::", "")))   # (k_icp_lat<4, 2> -> k_icp_lat)
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


# FETCH_SIZE is tallied at half the bytes for wide coalesced streaming reads (16 B and 4 B per lane alike) but in FULL for reads
# that a QUAD of lanes makes of 64 contiguous bytes at a scattered place (profiles/r04_fetch_calibration.txt, tools/fetch_calib.hip:
# 0.500 / 0.500 / 1.002 / 1.501 of the bytes read for stream16 / stream4 / quad64 aligned / quad64 at 16-byte alignment, where a
# segment straddles two 64-byte sectors half of the time).  The centroid kernels gather their points that way: factor 1 for them
# (rounds 1-3 doubled every kernel's FETCH_SIZE, which overstated these two by the size of their gather).
FETCH_FACTOR = {"k_voxel_centroid_runs": 1.0, "k_voxel_centroid": 1.0, "k_voxel_centroid_lanes": 1.0}   # (lanes: 16 scattered bytes per lane - tallied in full like the quads' 64)

fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 bench.py --steps 1 "
               "--warmup 0 --no-cpu-baseline --no-latency --no-verify; counters are in KB; on gfx950 FETCH_SIZE reads exactly half of a wide coalesced "
               "read (k_crop_fused streams the 1,258,291,200 B of input exactly once and shows ~617,000 KB; round 1 calibrated the same way on k_crop_count), so hbm_bytes = (2*FETCH_SIZE + "
               "WRITE_SIZE)*1024 - except for the kernels that gather by quads of lanes (fetch_factor 1: tools/fetch_calib.hip, profiles/r04_fetch_calibration.txt)",
       "workload": "256 frames x 307200 points, default bench config", "kernels": {}}
for k in sorted(fetch, key=lambda k: -(FETCH_FACTOR.get(k, 2.0) * fetch[k] + write.get(k, 0))):
    if not k.startswith("k_"):
        continue
    ff = FETCH_FACTOR.get(k, 2.0)
    out["kernels"][k] = {"dispatches": nf[k], "FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write.get(k, 0.0), "fetch_factor": ff,
                         "hbm_bytes_per_dispatch": (ff * fetch[k] + write.get(k, 0.0)) * 1024 / nf[k]}
out["total_hbm_bytes_per_batch"] = sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for v in out["kernels"].values())
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_dispatch"] for k, v in out["kernels"].items()}))
