#!/bin/bash
# the admission gate of the persistent ICP launches (CUBOID_ICP_CONCURRENT) on the driver's 20-step invocation and on the default 300 steps
cd "$(dirname "$0")/.."
B="--no-legs --no-cpu-baseline --no-latency"
for rep in 1 2 3 4; do for k in 0 2 3; do
  CUBOID_ICP_CONCURRENT=$k python bench.py --gpus 1 --steps 20 --warmup 5 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gate %s 20 steps: %.0f frames/s verified %s' % (sys.argv[1], d['value'], d['verified']))" $k
done; done
for rep in 1 2; do for k in 0 2 3; do
  CUBOID_ICP_CONCURRENT=$k python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gate %s 300 steps: %.0f frames/s verified %s' % (sys.argv[1], d['value'], d['verified']))" $k
done; done
