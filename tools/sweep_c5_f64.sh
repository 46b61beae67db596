#!/bin/bash
# config 5 at 64 frames per batch, four batches in flight: clusters per workgroup, the big template's workgroup share, stream priorities
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=16
for lp in 1 0; do for cpw in 1 2; do for bw in 2 4 8; do
  CUBOID_ICP_LOWPRIO=$lp CUBOID_ICP_CPW=$cpw CUBOID_ICP_BIG_WEIGHT=$bw python bench.py --config 5 --frames 64 --inflight 4 --steps 10 --warmup 4 --no-latency --no-verify 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('lowprio $lp cpw $cpw bigw $bw: %.0f frames/s  %.2f ms/step  icp %.2f ms' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['icp']))"
done; done; done
