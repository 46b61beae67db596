#!/bin/bash
cd $GRAFT_REPO_ROOT
for rc in 1.0 1.5 2.0 3.0 4.0; do
  CUBOID_ICP_GRID_RC=$rc python bench.py --config 5 --inflight 3 --steps 24 --no-latency 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('rc $rc: %.0f frames/s  %.2f ms/step  icp %.2f ms verified %s' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['icp'], d['verified']))"
done
