#!/bin/bash
# Round profile: kernel-trace stats of the default bench, FETCH/WRITE PMC passes (separate runs), SQ counters.
# Outputs under gpurun_out/prof/ (copy what should be judged into profiles/).
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-latency --no-verify > $OUT/bench_under_rocprof.json 2> /tmp/kt.log
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-verify > /tmp/pmc_$c.log 2>&1
  echo "$c done"
done
python3 $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $OUT/pmc_traffic.json > /dev/null
: > $OUT/pmc_icp.txt
n=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  n=$((n+1))
  rm -rf /tmp/sq_$n && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/sq_$n -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-verify > /tmp/sq_$n.log 2>&1
  echo "== pass $n (--pmc $pass), bench.py --steps 1 --warmup 0 --no-latency --no-verify" >> $OUT/pmc_icp.txt
  python3 $R/tools/pmc_summary.py /tmp/sq_$n | grep -E "^k_" >> $OUT/pmc_icp.txt
  echo "sq pass $n done"
done
