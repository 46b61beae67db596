#!/bin/bash
# A/B of k_icp_lat's workgroup size on the default bench (300 steps, 7 in flight) -> stdout
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs"
for t in 64 128 256 512; do
  for rep in 1 2; do
    v=$(CUBOID_LAT_THREADS=$t timeout -k 10 120 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  icp alone %s' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms')))")
    echo "threads $t rep $rep: $v"
  done
done
