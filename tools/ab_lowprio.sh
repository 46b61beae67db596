#!/bin/bash
# stream priorities of the persistent ICP launches: CUBOID_ICP_LOWPRIO = 0 (one stream), 1 (ICP on low-priority streams), 2 (and the
# front end on a high-priority one); config 3 and config 5 (the latter also by batches in flight)
cd "$(dirname "$0")/.."
for rep in 1 2; do for lp in 0 1 2; do  # (1 = default)
  CUBOID_ICP_LOWPRIO=$lp python bench.py --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 3 prio $lp: %.0f frames/s verified %s' % (d['value'], d['verified']))"
  for inf in 4 6 8; do
  CUBOID_ICP_LOWPRIO=$lp python bench.py --config 5 --inflight $inf --no-latency 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 5 prio $lp inflight $inf: %.0f frames/s verified %s' % (d['value'], d['verified']))"
  done
done; done
