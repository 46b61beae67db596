#!/bin/bash
# repeated A/B on one box, 300 steps, three rounds interleaved:
#   prev   = k_icp.hip of the previous commit (no pre-packed key payload)
#   main   = the shipped library (two slots)
#   slots3 = three clusters in flight per workgroup (make VARIANT=slots3 FLAGS_EXTRA=-DCD_PIPE_SLOTS=3)
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --steps 300 --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%s: %.0f frames/s  icp launch %.2f ms verified %s' % (sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['verified']))" "$LABEL"; }
for rep in 1 2 3; do
  [ -f perception_amd/lib/variants/libprev.so ] && LABEL="prev   wg 256 inflight 4" CUBOID_HIP_LIB=perception_amd/lib/variants/libprev.so run --inflight 4
  LABEL="main   wg 256 inflight 3" run --inflight 3
  LABEL="main   wg 256 inflight 4" run --inflight 4
  LABEL="main   wg 256 inflight 5" run --inflight 5
  LABEL="main   wg 256 inflight 6" run --inflight 6
  LABEL="slots3 wg 192 inflight 4" CUBOID_HIP_LIB=perception_amd/lib/variants/libslots3.so CUBOID_ICP_MAX_WG=192 run --inflight 4
done
