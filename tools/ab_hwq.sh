#!/bin/bash
# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share one queue run in order.  With
# 5-6 batches in flight (one context = one main stream + two low-priority ICP streams each) that aliasing can serialise
# batches that have nothing to do with each other.  Throughput against the number of hardware queues, config 3 and config 5.
cd "$(dirname "$0")/.."
for rep in 1 2; do for q in 4 8 16 24; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 3 hwq $q: %.0f frames/s verified %s' % (d['value'], d['verified']))"
  for inf in 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --config 5 --inflight $inf --no-latency 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 5 hwq $q inflight $inf: %.0f frames/s verified %s' % (d['value'], d['verified']))"
  done
done; done
