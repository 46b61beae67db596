#!/bin/bash
# A/B of environment settings on the default bench (300 steps): tools/ab_env.sh "A=1 B=2" "A=2" ...  ("-" = no setting)
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs $AB_BENCH_ARGS"
for rep in 1 2; do
  for e in "$@"; do
    [ "$e" = "-" ] && e=""
    v=$(env $e timeout -k 10 120 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  icp in flight %.3f ms  stages %s' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms',0), {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}))")
    echo "[$e] rep $rep: $v"
  done
done
