#!/bin/bash
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
for rep in 1 2; do for lp in 1 0; do for bw in 1 2 3; do for inf in 3 4; do
  CUBOID_ICP_LOWPRIO=$lp CUBOID_ICP_BIG_WEIGHT=$bw python bench.py --config 5 --frames 64 --inflight $inf --steps 10 --warmup 4 --no-latency --no-verify 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('lowprio $lp bigw $bw inflight $inf: %.0f frames/s  %.2f ms/step  icp %.2f ms' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['icp']))"
done; done; done; done
