#!/bin/bash
# CUBOID_ICP_REACH = t2,t3,t4: per 64-query pass the grid walk also takes seed balls up to 2 / 3 / 4 cells wide when at least
# t2 / t3 / t4 lanes are in that band.  Throughput on config 3 and config 5, records verified against the serial pass.
cd "$(dirname "$0")/.."
for r in ${REACHES:-0,0,0 10,0,0 20,0,0 10,22,0 10,22,40 6,16,30 16,32,48 0,0,0}; do
  CUBOID_ICP_REACH=$r python bench.py --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 3 reach $r: %.0f frames/s  icp kernel %.2f ms  verified %s' % (d['value'], d['roofline']['avg_launch_ms'], d['verified']))"
  CUBOID_ICP_REACH=$r python bench.py --config 5 --no-latency 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 5 reach $r: %.0f frames/s  verified %s' % (d['value'], d['verified']))"
done
