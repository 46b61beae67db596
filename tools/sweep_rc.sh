#!/bin/bash
# sweep the grid-search radius knob of the ICP kernels (tuning only; results are bit-identical for every value)
for rc in ${RC:-0 0.5 1.0 1.5 2.0}; do
  echo -n "grid_rc=$rc "
  CUBOID_ICP_GRID_RC=$rc timeout -k 10 120 python bench.py --no-cpu-baseline --inflight 1 --steps 4 --warmup 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['stage_ms_per_step']['icp'])" || exit 1
done
