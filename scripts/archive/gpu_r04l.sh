#!/bin/bash
set -u
TAG=${1:-r04l}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for BT in 8 0; do
MV_RASTER_BATCH=$BT timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t -o run -- python $R/bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-extra-legs --profile-steps 0 > $OUT/timeline_b$BT.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_t/run_results.db 80 200 > $OUT/timeline_b$BT.txt 2>> $OUT/timeline_b$BT.log
rm -rf $OUT/db_t
done
head -70 $OUT/timeline_b8.txt
