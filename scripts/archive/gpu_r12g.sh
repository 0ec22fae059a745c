#!/bin/bash
# r12g: kernel trace of the device-fed Collect gym: how long the draw launches last beside the passes, how many copy launches
set -u
TAG=${1:-r12g}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for M in 1 0; do
MV_COLLECT_DEVICE_GEN=$M timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_$M -o run -- python $R/bench.py --scenario Collect --steps 800 --warmup 96 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/stats_$M.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db_$M/run_results.db > $OUT/kernel_stats_devgen_$M.csv 2>> $OUT/stats_$M.log
python $R/scripts/kernel_timeline.py $OUT/db_$M/run_results.db 80 40 > $OUT/timeline_devgen_$M.txt 2>/dev/null
rm -rf $OUT/db_$M
cat $OUT/kernel_stats_devgen_$M.csv | cut -c1-150
done
