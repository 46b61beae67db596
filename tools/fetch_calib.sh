#!/bin/bash
# FETCH_SIZE of tools/bin/fetch_calib's four access patterns (1 GiB each) -> gpurun_out/fetch_calib.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc && timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o p --output-format csv -- $R/tools/bin/fetch_calib > /tmp/fc.log 2>&1 || { tail -5 /tmp/fc.log; exit 1; }
python3 - <<PY > $R/gpurun_out/fetch_calib.txt
import csv, glob
f = glob.glob("/tmp/fc/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and r["Kernel_Name"].startswith("k_")]
print("# rocprofv3 --pmc FETCH_SIZE -- tools/bin/fetch_calib: every kernel reads 1 GiB = 1048576 KB exactly once (FETCH_SIZE is in KB)")
agg = {}
for r in rows:
    k = (r["Kernel_Name"].split("(")[0], r.get("Dispatch_Id", ""))
    agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
for (k, d), v in sorted(agg.items(), key=lambda t: int(t[0][1] or 0)):
    print("%-12s dispatch %s FETCH_SIZE %.0f KB = %.3f of the bytes read" % (k, d, v, v / 1048576.0))
PY
cat $R/gpurun_out/fetch_calib.txt
