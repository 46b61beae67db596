#!/bin/bash
# throughput A/B of library builds on one box: tools/ab_lib.sh <reps> lib1.so lib2.so ...   (bench.py defaults, 300 steps)
cd "$(dirname "$0")/.."
reps=$1; shift
for rep in $(seq $reps); do for lib in "$@"; do
  CUBOID_HIP_LIB=$lib python bench.py --no-latency --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('%-48s %.0f frames/s  icp launch %.2f ms verified %s' % (sys.argv[1], d['value'], d['roofline']['avg_launch_ms'], d['verified']))" $lib
done; done
