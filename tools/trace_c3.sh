#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of the default bench under load, reduced to a compact table:
# gpurun_out/c3_trace.csv = kernel,queue,start,end,wg,grid (ns) of a 40 ms window in the steady state
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt -o b --output-format csv -- python3 $R/bench.py --steps 120 --no-cpu-baseline --no-latency --no-verify --no-legs $TRACE_ARGS > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
tail -1 /tmp/kt.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('under the tracer: %.0f frames/s, %.3f ms per step' % (d['value'], d['ms_per_step']))"
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
lo, hi = t_end - 90_000_000, t_end - 50_000_000
with open("$R/gpurun_out/c3_trace.csv", "w") as o:
    o.write("kernel,queue,start,end,wg,grid\n")
    for r in rows:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if b < lo or a > hi: continue
        o.write("%s,%s,%d,%d,%s,%s\n" % (r["Kernel_Name"].split("(")[0].replace("cd::", "").replace("void ", ""), r.get("Queue_Id", ""), a - lo, b - lo, r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Grid_Size_X", r.get("Grid_Size", ""))))
print("window written")
PY
