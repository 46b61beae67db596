#!/bin/bash
# PMC passes over the bench workload (1 step), summarised per kernel.  usage: tools/pmc_icp.sh "<counters pass1>" "<counters pass2>" ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
n=0
for pass in "$@"; do
  n=$((n+1))
  rm -rf /tmp/pmc_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$n -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency --no-verify --no-legs > /tmp/pmc_$n.log 2>&1 || { tail -5 /tmp/pmc_$n.log; exit 1; }
  echo "== pass $n: $pass"
  python3 $R/tools/pmc_summary.py /tmp/pmc_$n > /tmp/pmc_$n.txt; grep -E "${KERNELS:-k_icp}" /tmp/pmc_$n.txt
done
