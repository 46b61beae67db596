#!/bin/bash
# tools/ab_overlap.sh with library variants: args "lib:shape" ...
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-latency --no-verify --no-legs --steps 100 --inflight 2"
for spec in "$@"; do
  v=${spec%%:*}; sh=${spec#*:}
  lib=""; [ "$v" != "-" ] && lib="CUBOID_HIP_LIB=$GRAFT_REPO_ROOT/perception_amd/lib/variants/lib$v.so"
  r=$(env $lib CUBOID_FRONT_CONCURRENT=1 CUBOID_ICP_CONCURRENT=1 CUBOID_LAT_SHAPE=$sh timeout -k 10 150 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f frames/s  %.3f ms/step  stages %s wave_ms %.0f' % (d['value'], d['ms_per_step'], {k: round(v,2) for k,v in d['stage_ms_per_step'].items()}, (d['roofline'].get('wave_time') or {}).get('wave_ms_per_batch',0)))")
  echo "lib $v shape $sh: $r"
done
