import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('cd::', '')
    by[name].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Grid_Size', ''), r.get('VGPR_Count', ''), r.get('LDS_Block_Size', '')))
for n, v in sorted(by.items(), key=lambda kv: -sum(d for _, d, *_ in kv[1])):
    ds = [d for _, d, *_ in v]
    print('%-28s calls %5d total %9.3f ms avg %8.1f us min %7.1f max %8.1f  grid %s vgpr %s lds %s' % (n, len(ds), sum(ds) / 1e6, sum(ds) / len(ds) / 1e3, min(ds) / 1e3, max(ds) / 1e3, v[0][2], v[0][3], v[0][4]))
it = by.get('k_icp_iter', [])
last = it[-int(sys.argv[2]):] if len(sys.argv) > 2 else it
print('k_icp_iter durations (us) of the last batch, in launch order:')
print(' '.join('%d' % (d / 1e3) for _, d, *_ in last))
so = by.get('k_icp_solve', [])
if so:
    print('k_icp_solve (us):', ' '.join('%d' % (d / 1e3) for _, d, *_ in so[-int(sys.argv[2]):]))
# gaps between consecutive kernels of the last batch
t0 = last[0][0]; t1 = last[-1][0] + last[-1][1]
busy = sum(d for s, d, *_ in rows_ if True) if False else None
span = [r for r in rows if t0 <= int(r['Start_Timestamp']) <= t1]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in span)
print('icp span %.3f ms, kernel busy %.3f ms, kernels %d' % ((t1 - t0) / 1e6, busy / 1e6, len(span)))
