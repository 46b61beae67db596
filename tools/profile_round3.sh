#!/bin/bash
# Round-3 profile set -> gpurun_out/prof/ (copied into profiles/r03_* afterwards):
#   tools/profile_round.sh   kernel stats of the default bench under rocprofv3, FETCH/WRITE PMC passes, four SQ passes
#   + kernel stats of bench.py --config 5 and of the strictly serial bench batch, the big-template timing, the executed-work probe
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
$R/tools/profile_round.sh
echo "round profile done"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt5 && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o b --output-format csv -- python3 $R/bench.py --config 5 --no-latency --no-verify > $OUT/bench_config5_under_rocprof.json 2> /tmp/kt5.log
cp $(find /tmp/kt5 -name '*kernel_stats.csv' | head -1) $OUT/config5_kernel_stats.csv
echo "config 5 stats done"
cd $R
tools/serial_kernel_ms.sh > $OUT/serial_kernel_ms.txt 2>&1 && cp gpurun_out/serial_kernel_stats.csv $OUT/serial_kernel_stats.csv
echo "serial stats done"
python3 tools/big_template_ms.py 1 32 256 > $OUT/big_template_ms.txt 2>&1
echo "big template done"
CUBOID_HIP_LIB=perception_amd/lib/variants/libstats.so python3 tools/probe_icp_work.py $OUT/icp_work.json > /dev/null 2>&1 || echo "work probe failed"
python3 bench.py > $OUT/bench_line.json 2> /dev/null
python3 bench.py --config 5 > $OUT/bench_line_config5.json 2> /dev/null
echo "bench lines done"
