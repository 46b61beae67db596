#!/bin/bash
# Round-5 profile set -> gpurun_out/prof5/ (copied into profiles/r05_* afterwards).  As tools/profile_round4.sh; the dominant
# ICP kernel is now k_icp_lat (closed-form nearest neighbour, one workgroup per cluster), its counters are collected on the
# launch of a one-batch run (bench.py --steps 1 --warmup 0).  $1 = "quick": kernel stats only (in flight and serial).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-latency --no-verify --no-legs"
rm -rf /tmp/kt && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $R/bench.py $B > $OUT/bench_under_rocprof.json 2> /tmp/kt.log
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
echo "stats done"
cd $R
tools/serial_kernel_ms.sh > $OUT/serial_kernel_ms.txt 2>&1 && cp gpurun_out/serial_kernel_stats.csv $OUT/serial_kernel_stats.csv
echo "serial stats done"
[ "$1" = quick ] && exit 0
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $B > /tmp/pmc_$c.log 2>&1
  echo "$c done"
done
python3 $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/pmc_traffic.json > /dev/null
: > $OUT/pmc_icp.txt
: > $OUT/pmc_icp_inflight_shape.txt
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_WAVES")
# two sets: the launch shape of a call that has the GPU to itself (1 cluster x 8 waves per workgroup for the 522 clusters of the bench batch: what the one-batch run picks by
# itself) and the shape the launches of the timed region have (4 clusters x 2 waves, forced here; counters serialise the kernels,
# so "under load" can only mean the shape)
for set in "excl:" "inflight:CUBOID_LAT_SHAPE=4,2"; do
  tag=${set%%:*}; envs=${set#*:}
  out=$OUT/pmc_icp.txt; [ "$tag" = inflight ] && out=$OUT/pmc_icp_inflight_shape.txt
  n=0
  for pass in "${PASSES[@]}"; do
    n=$((n+1))
    rm -rf /tmp/sq_$n
    if [ -n "$envs" ]; then export $envs; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/sq_$n -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $B > /tmp/sq_$n.log 2>&1
    unset CUBOID_LAT_SHAPE
    echo "== pass $n (--pmc $pass), ${envs:-default shape} bench.py --steps 1 --warmup 0 $B" >> $out
    python3 $R/tools/pmc_summary.py /tmp/sq_$n | grep -E "^k_" >> $out
    echo "$tag sq pass $n done"
  done
done
