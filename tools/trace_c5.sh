#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace) of config 5 under load, reduced to a compact table for tools/analyze_trace.py:
# gpurun_out/c5_trace.csv = kernel, queue, start, end (ns)
R=$GRAFT_REPO_ROOT
INF=${1:-6}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt -o b --output-format csv -- python3 $R/bench.py --config 5 --inflight $INF --steps 24 --warmup 6 --no-latency --no-verify > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
tail -1 /tmp/kt.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('under the tracer: %.0f frames/s' % d['value'])"
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "dispatches; columns", list(rows[0].keys()))
with open("$R/gpurun_out/c5_trace.csv", "w") as o:
    o.write("kernel,queue,stream,start,end,wg,grid\n")
    for r in rows:
        o.write("%s,%s,%s,%s,%s,%s,%s\n" % (r["Kernel_Name"].split("(")[0].replace("cd::", "").replace("void ", ""), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Start_Timestamp"], r["End_Timestamp"], r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Grid_Size_X", r.get("Grid_Size", ""))))
PY
