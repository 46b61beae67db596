"""Per (stream, workgroup size) counter values of tools/bin/valu_calib under rocprofv3 --pmc: the timed dispatch of each
(the one with the most work), not the warm-up."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = collections.defaultdict(dict)
meta = {}
for r in rows:
    d = r.get('Dispatch_Id') or r.get('Correlation_Id')
    disp[d][r['Counter_Name']] = disp[d].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    meta[d] = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Workgroup_Size', '?'))
best = {}
for d, c in disp.items():
    key = meta[d]
    w = sum(c.values())
    if key not in best or w > best[key][0]:
        best[key] = (w, c)
names = ['fma_indep', 'fma_dep', 'add_indep', 'pk_indep', 'key_min', 'lds_chase', 'mix_search']
for (k, wg) in sorted(best, key=lambda t: (int(t[1]) if t[1].isdigit() else 0, t[0])):
    c = best[(k, wg)][1]
    try:
        label = names[int(k.split('<')[1].split('>')[0])]
    except Exception:
        label = k
    print(label, 'wg', wg, {n: ('%.5g' % v) for n, v in sorted(c.items())})
