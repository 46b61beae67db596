#!/bin/bash
# k_icp_lat launch shapes (clusters per workgroup, waves per cluster[, clusters per slot]) on the default bench, in flight
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs"
for sh in "$@"; do
  v=$(CUBOID_LAT_SHAPE=$sh timeout -k 10 120 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  icp in flight %.3f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms',0)))")
  echo "shape $sh: $v"
done
