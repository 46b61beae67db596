#!/bin/bash
# generic A/B on the default bench: each argument is "ENV=.. ENV=.. -- bench args" ; prints frames/s
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs"
for spec in "$@"; do
  e="${spec%%--*}"; a="${spec#*--}"; [ "$a" = "$spec" ] && a=""
  v=$(env $e timeout -k 10 150 python3 bench.py $B $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  icp in flight %.3f ms  device_total %.2f' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms',0), d['stage_ms_per_step']['device_total']))")
  echo "[$spec]: $v"
done
