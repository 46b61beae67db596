#!/bin/bash
# config-3 throughput against the persistent ICP grid size and the number of batches in flight
cd "$(dirname "$0")/.."
for wg in ${WGS:-128 160 192 224 256}; do for inf in ${INFL:-2 3 4}; do
  CUBOID_ICP_MAX_WG=$wg python bench.py --inflight $inf --steps ${STEPS:-150} --no-latency --no-verify --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('max_wg $wg inflight $inf: %.0f frames/s  %.3f ms/step  icp launch %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done; done
