#!/bin/bash
# config-5 throughput against clusters-per-workgroup, batches in flight and the big template's workgroup share
cd "$(dirname "$0")/.."
for cpw in 1 2 3; do for inf in 3 6; do for bw in 2 4; do
  CUBOID_ICP_CPW=$cpw CUBOID_ICP_BIG_WEIGHT=$bw python bench.py --config 5 --inflight $inf --steps 30 --no-latency --no-verify 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('cpw $cpw inflight $inf bigw $bw: %.0f frames/s  %.2f ms/step  icp %.2f ms' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['icp']))"
done; done; done
