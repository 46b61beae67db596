#!/bin/bash
# host-side cost of the HIP runtime calls of the default bench (rocprofv3 --hip-trace --stats) -> stdout
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht
timeout -k 10 400 rocprofv3 --hip-trace --stats -d /tmp/ht -o h --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-latency --no-verify --no-legs --steps 150 > /tmp/ht.json 2> /tmp/ht.log || { tail -5 /tmp/ht.log; exit 1; }
tail -1 /tmp/ht.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('under the tracer: %.0f frames/s, %.3f ms per step, %d steps' % (d['value'], d['ms_per_step'], d['steps']))"
python3 - <<PY
import csv, glob
f = [x for x in glob.glob("/tmp/ht/**/*stats.csv", recursive=True) if "hip" in x][0]
rows = list(csv.DictReader(open(f)))
print(f)
for r in rows[:18]:
    print("%-34s calls %7s  total %9.1f ms  avg %8.1f us  %5s %%" % (r["Name"][:34], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
python3 - <<PY
import csv, glob, collections
f = [x for x in glob.glob("/tmp/ht/**/*hip_api_trace.csv", recursive=True)][0]
d = collections.defaultdict(list)
tids = collections.Counter()
for r in csv.DictReader(open(f)):
    d[r["Function"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tids[r.get("Thread_Id", "")] += 1
import statistics
for k in ("hipMemcpyAsync", "hipMemcpy2DAsync", "hipLaunchKernel", "hipMemsetAsync", "hipEventRecord", "hipStreamSynchronize"):
    v = sorted(d.get(k, []))
    if v:
        print("%-22s n %6d  p10 %8.1f  p50 %8.1f  p90 %8.1f  p99 %8.1f  max %9.1f us  sum %8.1f ms" % (k, len(v), v[len(v)//10], v[len(v)//2], v[len(v)*9//10], v[len(v)*99//100], v[-1], sum(v)/1e3))
print("threads:", len(tids))
PY
