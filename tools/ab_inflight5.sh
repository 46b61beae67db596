#!/bin/bash
# batches in flight on the default bench (300 steps) -> stdout; extra environment in $AB_ENV
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs"
for n in "$@"; do
  v=$(env $AB_ENV timeout -k 10 150 python3 bench.py $B --inflight $n 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  icp in flight %.3f ms  stages %s' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms',0), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}))")
  echo "inflight $n: $v"
done
