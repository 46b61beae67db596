#!/bin/bash
# long host-side HIP calls of the driver's shape (--steps 20 --warmup 5): every call over 300 us in the last 60 ms of API activity
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht
timeout -k 10 400 rocprofv3 --hip-trace -d /tmp/ht -o h --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-verify --no-legs > /tmp/ht.json 2> /tmp/ht.log || { tail -5 /tmp/ht.log; exit 1; }
tail -1 /tmp/ht.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('under the tracer: %.0f frames/s, %.3f ms per step' % (d['value'], d['ms_per_step']))"
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/ht/**/*hip_api_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
launches = [r for r in rows if r[2] == "hipLaunchKernel"]
tend = launches[-1][1]
t0 = tend - 70_000_000
print("calls over 300 us in the last 70 ms before the last kernel launch (ms relative to that window's start):")
for a, b, fn, tid in rows:
    if a >= t0 and a <= tend and b - a > 300_000 and fn != "hipStreamSynchronize":
        print("  %8.2f ms  %-28s %9.1f us  thread %s" % ((a - t0) / 1e6, fn, (b - a) / 1e3, tid))
# gaps: per thread, time between consecutive API calls > 2 ms (the thread was elsewhere: Python, locks)
import collections
by = collections.defaultdict(list)
for r in rows:
    if r[0] >= t0 and r[0] <= tend: by[r[3]].append(r)
for tid, v in by.items():
    for x, y in zip(v, v[1:]):
        if y[0] - x[1] > 2_000_000:
            print("  thread %s idle between %s (end %.2f ms) and %s (start %.2f ms)" % (tid, x[2], (x[1] - t0) / 1e6, y[2], (y[0] - t0) / 1e6))
PY
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/ht/**/*hip_api_trace.csv", recursive=True)[0]
rd = list(csv.DictReader(open(f)))
print("columns:", list(rd[0].keys()))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")) for r in rd]
rows.sort()
launches = [r for r in rows if r[2] == "hipLaunchKernel"]
tend = launches[-1][1]
t0 = tend - 70_000_000
long_ = [r for r in rows if r[0] >= t0 and r[1] - r[0] > 3_000_000 and r[2] == "hipMemcpyAsync"]
for L in long_[:2]:
    tid = L[3]
    seq = [r for r in rows if r[3] == tid and L[0] - 3_000_000 <= r[0] <= L[1] + 1_000_000]
    print("thread", tid, "around the long call at %.2f ms:" % ((L[0] - t0) / 1e6))
    for a, b, fn, _ in seq[-40:]:
        print("    %8.3f ms %-26s %8.1f us" % ((a - t0) / 1e6, fn, (b - a) / 1e3))
print("allocation calls in the window:")
for a, b, fn, tid in rows:
    if a >= t0 and a <= tend and fn in ("hipFree", "hipMalloc", "hipHostMalloc", "hipHostFree", "hipMemcpy", "hipDeviceSynchronize", "hipMemcpyWithStream", "hipStreamCreateWithPriority"):
        print("  %8.2f ms %-24s %8.1f us thread %s" % ((a - t0) / 1e6, fn, (b - a) / 1e3, tid))
PY
python3 $R/tools/hip_api_window.py
