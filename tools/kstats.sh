#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python script of this repo: tools/kstats.sh tools/probe_latency.py [args]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $R/"$@" > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-36s calls %6s avg %9.1f us  %5s %%" % (r["Name"].split("(")[0].replace("cd::", "")[:36], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
