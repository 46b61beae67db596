#!/bin/bash
# shipped library: clusters in flight per workgroup (CUBOID_ICP_SLOTS) x persistent grid (CUBOID_ICP_MAX_WG) x batches in flight
cd "$(dirname "$0")/.."
for inf in ${INFL:-5 6}; do for sl in ${SLOTS:-3 4}; do for wg in ${WGS:-96 128 160 192}; do
  CUBOID_ICP_SLOTS=$sl CUBOID_ICP_MAX_WG=$wg CUBOID_ICP_CPW=1 python bench.py --inflight $inf --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('slots $sl wg $wg inflight $inf: %.0f frames/s  icp launch %.2f ms verified %s' % (d['value'], d['roofline']['avg_launch_ms'], d['verified']))"
done; done
python bench.py --inflight $inf --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('default (by regime) inflight $inf: %.0f frames/s  icp launch %.2f ms verified %s' % (d['value'], d['roofline']['avg_launch_ms'], d['verified']))"
done
