import csv, glob, collections
f = glob.glob("/tmp/ht/**/*hip_api_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
launches = [r for r in rows if r[2] == "hipLaunchKernel"]
tend = launches[-1][1]
t0 = tend - 70_000_000
long_ = [r for r in rows if r[0] >= t0 and r[1] - r[0] > 3_000_000 and r[2] == "hipMemcpyAsync"]
L = long_[0]
print("window %.2f .. %.2f ms: every call of every thread that overlaps it and lasts > 50 us, plus all calls of threads other than the five blocked ones" % ((L[0]-t0)/1e6, (L[1]-t0)/1e6))
blocked = {r[3] for r in long_}
for a, b, fn, tid in rows:
    if b >= L[0] - 1_500_000 and a <= L[1] + 200_000:
        if (b - a > 50_000) or tid not in blocked:
            print("  %8.3f -> %8.3f ms  %-28s %9.1f us thread %s" % ((a - t0) / 1e6, (b - t0) / 1e6, fn, (b - a) / 1e3, tid))
