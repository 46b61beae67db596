#!/bin/bash
# config 5: frames per batch x batches in flight (the persistent ICP launches of a batch last as long as their longest cluster,
# ~15 ms, whatever the batch size: a bigger batch puts more workgroups under that tail)
cd "$(dirname "$0")/.."
for f in ${FRAMES:-8 16 32 64}; do for inf in ${INFL:-2 3 4 6}; do
  python bench.py --config 5 --frames $f --inflight $inf --steps $((320 / f)) --warmup 4 --no-latency 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 5 frames $f inflight $inf: %.0f frames/s  %.2f ms/step  verified %s' % (d['value'], d['ms_per_step'], d['verified']))"
done; done
