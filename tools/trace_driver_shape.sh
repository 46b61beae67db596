#!/bin/bash
# kernel timeline of the driver's own invocation (--steps 20 --warmup 5): gpurun_out/driver_shape_trace.csv = kernel;queue;start;end (us from the first kernel of the timed region)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/kt -o b --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-latency --no-verify --no-legs $TRACE_ARGS > /tmp/kt.log 2>&1 || { tail -5 /tmp/kt.log; exit 1; }
tail -1 /tmp/kt.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('under the tracer: %.0f frames/s, %.3f ms per step' % (d['value'], d['ms_per_step']))"
python3 - <<PY
import csv, glob, re
f = glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return re.sub(r"<.*>$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].split("(")[0].replace("cd::", "")))
crops = [r for r in rows if nm(r) == "k_crop_runs"]
# the timed region: the last 20 crops
t0 = int(crops[-20]["Start_Timestamp"]) - 200000
with open("$R/gpurun_out/driver_shape_trace.csv", "w") as o:
    o.write("kernel;queue;start_us;end_us\n")
    for r in rows:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if b < t0: continue
        o.write("%s;%s;%.1f;%.1f\n" % (nm(r), r.get("Queue_Id", ""), (a - t0) / 1e3, (b - t0) / 1e3))
print("written")
PY
