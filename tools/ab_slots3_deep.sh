#!/bin/bash
# three pipeline slots (make VARIANT=slots3 FLAGS_EXTRA=-DCD_PIPE_SLOTS=3) against the shipped two, with the "any free context"
# submission and deeper pipelines; repeated, interleaved, 300 steps
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --steps 300 --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%s: %.0f frames/s  icp launch %.2f ms verified %s' % (sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['verified']))" "$LABEL"; }
for rep in 1 2 3; do
  for inf in 5 7; do
    LABEL="main   wg 256 inflight $inf" run --inflight $inf
    for wg in 160 176 192; do
      LABEL="slots3 wg $wg inflight $inf" CUBOID_HIP_LIB=perception_amd/lib/variants/libslots3.so CUBOID_ICP_MAX_WG=$wg run --inflight $inf
    done
  done
done
