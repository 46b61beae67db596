#!/bin/bash
# How full are the CUs under load?  A persistent ICP workgroup holds a whole CU; a -DCD_TIMERS build sums the workgroups' busy
# time, bench.py divides by wall time x 256 CUs ("cu_fill_debug").  (tools/build_variant.sh timers k_icp.hip -DCD_TIMERS first.)
cd "$(dirname "$0")/.."
export CUBOID_HIP_LIB=$PWD/perception_amd/lib/variants/libtimers.so
python bench.py --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 3: %.0f frames/s' % d['value'], d['cu_fill_debug'])"
for inf in 3 6; do
python bench.py --config 5 --inflight $inf --no-latency 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('config 5 inflight $inf: %.0f frames/s' % d['value'], d['cu_fill_debug'])"
done
