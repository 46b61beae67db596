#!/bin/bash
# pipeline slots (library variants libslotsN.so: make VARIANT=slotsN FLAGS_EXTRA=-DCD_PIPE_SLOTS=N) x persistent grid size x batches in flight
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --steps 300 --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%s: %.0f frames/s  icp launch %.2f ms verified %s' % (sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['verified']))" "$LABEL"; }
for rep in 1 2; do for inf in ${INFL:-7 9}; do
  LABEL="slots 2 wg 256 inflight $inf" run --inflight $inf
  for sl in ${SLOTS:-3 4}; do for wg in ${WGS:-112 128 144 160}; do
    LABEL="slots $sl wg $wg inflight $inf" CUBOID_HIP_LIB=perception_amd/lib/variants/libslots$sl.so CUBOID_ICP_MAX_WG=$wg run --inflight $inf
  done; done
done; done
