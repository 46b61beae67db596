#!/bin/bash
# Is the pipeline's ceiling on the host side (locks of the HIP runtime shared by the context threads of ONE process)?
# Same number of batches in flight, as one process and split over two / three processes that share the GPU.
# usage: tools/two_procs.sh   (prints frames/s of every process and their sum; the processes' timed regions overlap for
# most of their length: 3000 steps of ~2 ms against a start skew of well under a second)
B="--no-cpu-baseline --no-latency --no-verify --no-legs"
run() {  # n_procs inflight_each steps
  local n=$1 k=$2 s=$3 pids=()
  for i in $(seq 1 $n); do
    timeout -k 10 300 python3 bench.py $B --steps $s --inflight $k > /tmp/two_procs_$i.json 2>/tmp/two_procs_$i.err &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p || echo "a process failed"; done
  python3 - $n $k <<'PY'
import json, sys
n, k = int(sys.argv[1]), int(sys.argv[2])
vs = []
for i in range(1, n + 1):
    d = json.loads(open('/tmp/two_procs_%d.json' % i).read().strip().splitlines()[-1])
    vs.append(d['value'])
print('%d process(es) x %d in flight: %s -> sum %.0f frames/s' % (n, k, ' + '.join('%.0f' % v for v in vs), sum(vs)))
PY
}
run 1 6 3000
run 2 3 1500
run 3 2 1000
run 1 8 3000
run 2 4 1500
