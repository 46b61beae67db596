#!/bin/bash
# shader clock and power while the default bench runs (rocm-smi polled twice a second) -> stdout
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs --steps 1500"
( env $PROBE_ENV timeout -k 10 150 python3 bench.py $B > /tmp/bench_clk.json 2>/dev/null ) &
BP=$!
sleep 14
for i in $(seq 1 12); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/bench_clk.json').read().strip().splitlines()[-1]); print('%.0f frames/s' % d['value'])"
