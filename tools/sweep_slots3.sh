#!/bin/bash
# three clusters in flight per workgroup (make -C perception_amd/csrc VARIANT=slots3 FLAGS_EXTRA=-DCD_PIPE_SLOTS=3) against the
# shipped two, by persistent grid size and batches in flight; and the effect of a batch twice as large
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --steps 150 --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%s: %.0f frames/s  %.3f ms/step  icp launch %.2f ms verified %s' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['verified']))" "$LABEL"; }
for wg in 176 208 256; do for inf in 3 4; do
  LABEL="slots 3 max_wg $wg inflight $inf" CUBOID_HIP_LIB=perception_amd/lib/variants/libslots3.so CUBOID_ICP_MAX_WG=$wg run --inflight $inf
done; done
LABEL="slots 2 max_wg 256 inflight 3" run --inflight 3
LABEL="slots 2 frames 512 inflight 2" run --inflight 2 --frames 512
LABEL="slots 2 frames 512 inflight 3" run --inflight 3 --frames 512
LABEL="slots 3 frames 512 inflight 3" CUBOID_HIP_LIB=perception_amd/lib/variants/libslots3.so run --inflight 3 --frames 512
