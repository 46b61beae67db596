#!/bin/bash
# Per-kernel durations of the bench batch run strictly serially (one context, nothing else on the GPU):
# rocprofv3 kernel-trace of tools/icp_ms.py -> gpurun_out/serial_kernel_stats.csv
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export ICP_MS_NOCHECK=1
rm -rf /tmp/ks && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s --output-format csv -- python3 $R/tools/icp_ms.py 256 8 > $R/gpurun_out/serial_icp_ms.txt 2> /tmp/ks.log
cp $(find /tmp/ks -name '*kernel_stats.csv' | head -1) $R/gpurun_out/serial_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/serial_kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"].split("(")[0].replace("cd::","").replace("void ","")   # (templated kernels are listed as "void cd::k_...<...>")
    per_batch=float(r["TotalDurationNs"])/8/1e6
    tot+=per_batch if n.startswith("k_") else 0
    print("%-28s calls %5s  avg %9.1f us  per batch %7.3f ms" % (n, r["Calls"], float(r["AverageNs"])/1e3, per_batch))
print("sum of k_* per batch: %.3f ms" % tot)
PY
