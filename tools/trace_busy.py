"""GPU busy fraction from a rocprofv3 kernel trace: union of kernel intervals over the span of the last N batches."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('cd::', '')) for r in csv.DictReader(open(f))]
rows.sort()
icp = [r for r in rows if r[2].startswith('k_icp_pipe')]
t0, t1 = icp[len(icp) // 2][0], icp[-1][1]      # second half of the run: steady state
sel = [(max(a, t0), min(b, t1)) for a, b, _ in rows if b > t0 and a < t1]
sel.sort()
busy, cur_a, cur_b = 0, None, None
for a, b in sel:
    if cur_b is None or a > cur_b:
        if cur_b is not None: busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
n_icp = sum(1 for r in icp if r[0] >= t0)
print('span %.2f ms, %d ICP launches -> %.2f ms per batch; some kernel running %.1f %% of the time' % ((t1 - t0) / 1e6, n_icp, (t1 - t0) / 1e6 / max(n_icp, 1), 100.0 * busy / (t1 - t0)))
ov = 0
icps = sorted((r[0], r[1]) for r in icp if r[0] >= t0)
for (a0, b0), (a1, b1) in zip(icps, icps[1:]):
    ov += max(0, b0 - a1)
print('ICP launches overlapping each other: %.2f ms in total' % (ov / 1e6))
