#!/bin/bash
# the round-end driver's own invocation (python3 bench.py --gpus 1 --steps 20 --warmup 5), a few times, with extra args / env -> frames/s
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for spec in "$@"; do
    e="${spec%%--*}"; a="${spec#*--}"; [ "$a" = "$spec" ] && a=""
    v=$(env $e timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-latency $a 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  inflight %s verified %s device_total %.2f' % (d['value'], d['ms_per_step'], d['config']['batches_in_flight'], d['verified'], d['stage_ms_per_step']['device_total']))")
    echo "[$spec] run $i: $v"
  done
done
