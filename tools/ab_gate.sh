#!/bin/bash
# admission gate of the persistent ICP launches (CUBOID_ICP_CONCURRENT = K contexts inside an ICP launch at a time) against
# batches in flight, config 5 and config 3
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=${HWQ:-16}
for k in 0 3 4 5; do for inf in 6 8 12; do
  CUBOID_ICP_CONCURRENT=$k python bench.py --config 5 --inflight $inf --no-latency 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 5 gate $k inflight $inf: %.0f frames/s verified %s' % (d['value'], d['verified']))"
done; done
for k in 0 2 3 4; do for inf in 5 7; do
  CUBOID_ICP_CONCURRENT=$k python bench.py --inflight $inf --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 3 gate $k inflight $inf: %.0f frames/s verified %s' % (d['value'], d['verified']))"
done; done
