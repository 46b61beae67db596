#!/bin/bash
# VALU calibration: tools/bin/valu_calib plain, then under the same --pmc passes as tools/pmc_icp.sh.
# usage (on the GPU box): bash tools/valu_calib.sh   -> gpurun_out/calib/{plain.txt,pmc.txt}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $R/tools/bin/valu_calib 2000 > $OUT/plain.txt 2>&1 || { tail -5 $OUT/plain.txt; exit 1; }
: > $OUT/pmc.txt
n=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
            "SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$((n+1))
  rm -rf /tmp/calib_$n
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass -d /tmp/calib_$n -o p --output-format csv -- $R/tools/bin/valu_calib 2000 > /tmp/calib_$n.log 2>&1 || { tail -5 /tmp/calib_$n.log; exit 1; }
  echo "== pass $n: $pass" >> $OUT/pmc.txt
  python3 $R/tools/valu_calib_summary.py /tmp/calib_$n >> $OUT/pmc.txt
done
cat $OUT/plain.txt
