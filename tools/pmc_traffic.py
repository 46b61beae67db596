"""Merge two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/r01_pmc_traffic.json.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Counters are in KB; on gfx950 FETCH_SIZE reads half of a wide coalesced read (MI355X_MICROARCH.md,
calibrated on k_crop_count which streams exactly frames*points*16 B), so hbm = (2*FETCH + WRITE)*1024.
"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("cd::", "")
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) -- python3 bench.py --steps 1 "
               "--warmup 0 --no-cpu-baseline --no-latency --no-verify; counters are in KB; on gfx950 FETCH_SIZE reads exactly half of a wide coalesced "
               "read (k_crop_fused streams the 1,258,291,200 B of input exactly once and shows ~617,000 KB; round 1 calibrated the same way on k_crop_count), so hbm_bytes = (2*FETCH_SIZE + "
               "WRITE_SIZE)*1024",
       "workload": "256 frames x 307200 points, default bench config", "kernels": {}}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    if not k.startswith("k_"):
        continue
    out["kernels"][k] = {"dispatches": nf[k], "FETCH_SIZE_KB": fetch[k], "WRITE_SIZE_KB": write.get(k, 0.0),
                         "hbm_bytes_per_dispatch": (2 * fetch[k] + write.get(k, 0.0)) * 1024 / nf[k]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_dispatch"] for k, v in out["kernels"].items()}))
