#!/bin/bash
# SQ counter passes over one step of config 5 (far-heavy ICP): tools/pmc_c5.sh  -> gpurun_out/pmc_c5.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/pmc_c5.txt
n=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"; do
  n=$((n+1))
  rm -rf /tmp/c5_$n && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/c5_$n -o p --output-format csv -- python3 $R/bench.py --config 5 --inflight 1 --steps 1 --warmup 0 --no-latency --no-verify > /tmp/c5_$n.log 2>&1
  echo "== pass $n (--pmc $pass), bench.py --config 5 --inflight 1 --steps 1 --warmup 0" >> $R/gpurun_out/pmc_c5.txt
  python3 $R/tools/pmc_summary.py /tmp/c5_$n | grep -E "^k_icp" >> $R/gpurun_out/pmc_c5.txt
done
cat $R/gpurun_out/pmc_c5.txt
