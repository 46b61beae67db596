#!/bin/bash
# usage (on the GPU box): tools/variant_kernel_ms.sh <kernel-name-substring> lib1.so [lib2.so ...]
# Serial per-batch time of one kernel for several builds of the library (rocprofv3 kernel trace of tools/icp_ms.py).
R=$GRAFT_REPO_ROOT
K=$1; shift
cd /tmp && export TMPDIR=/tmp
export ICP_MS_NOCHECK=1
for lib in "$@"; do
  export CUBOID_HIP_LIB=$R/$lib
  rm -rf /tmp/kv && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kv -o s --output-format csv -- python3 $R/tools/icp_ms.py 256 6 > /tmp/kv.out 2> /tmp/kv.log || { echo "$lib FAILED"; tail -5 /tmp/kv.log; exit 1; }
  echo "== $lib: $(tail -1 /tmp/kv.out)"
  python3 - "$K" $(find /tmp/kv -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[2])):
    if re.search(sys.argv[1], r["Name"]):
        print("   %-28s calls %s avg %.1f us min %.1f us" % (r["Name"].replace("cd::", "").split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
