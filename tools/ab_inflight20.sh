#!/bin/bash
# the driver's invocation (--steps 20 --warmup 5) against the number of batches in flight: the timed region then includes the fill
# and the drain of the pipeline
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for inf in 3 4 5 6 8; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --inflight $inf --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('steps 20 inflight $inf: %.0f frames/s  %.3f ms/step verified %s' % (d['value'], d['ms_per_step'], d['verified']))"
done; done
