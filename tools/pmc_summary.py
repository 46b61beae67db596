import csv, sys, glob, collections, re
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    name = re.sub(r'<.*>$', '', re.sub(r'^void\s+', '', r['Kernel_Name'].split('(')[0].replace('cd::', '')))   # (k_icp_lat<4, 2> -> k_icp_lat)
    agg[name][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(name, r['Counter_Name'])] += 1
for name in sorted(agg, key=lambda n: -agg[n].get('SQ_WAVE_CYCLES', agg[n].get('SQ_WAVES', 0))):
    print(name, {k: ('%.4g' % v) for k, v in sorted(agg[name].items())}, 'dispatches', max(cnt[(name, k)] for k in agg[name]))
