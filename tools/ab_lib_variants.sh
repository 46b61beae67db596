#!/bin/bash
# library variants (perception_amd/lib/variants/lib<name>.so; "-" = the default library) x CUBOID_LAT_SHAPE: serial ICP time and in-flight throughput
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs"
for v in "$@"; do
  lib=""; [ "$v" != "-" ] && lib="CUBOID_HIP_LIB=$GRAFT_REPO_ROOT/perception_amd/lib/variants/lib$v.so"
  for sh in $AB_SHAPES; do
    a=$(env $lib ICP_MS_NOCHECK=1 CUBOID_LAT_SHAPE=$sh timeout -k 10 100 python3 tools/icp_ms.py 256 6 2>/dev/null | tail -1 | cut -d'|' -f1)
    b=$(env $lib CUBOID_LAT_SHAPE=$sh timeout -k 10 120 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  icp in flight %.3f ms' % (d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms',0)))")
    echo "lib $v shape $sh: $a | $b"
  done
done
