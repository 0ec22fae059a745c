#!/bin/bash
# r12j: what slows the observation launches beside a draw launch (r12h: 1014 -> 1105 us)?  MV_DRAW_EXPERIMENT=2: every draw launch stays resident for 20 ms with ONE sleeping wave after its work is done
set -u
TAG=${1:-r12j}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for X in 0 2; do
MV_DRAW_EXPERIMENT=$X MV_COLLECT_DEVICE_GEN=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/db_$X -o run -- python $R/bench.py --scenario Collect --steps 4000 --warmup 96 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/stats_$X.log 2>&1
python $R/scripts/draw_overlap.py $OUT/db_$X/run_results.db > $OUT/draw_overlap_experiment_$X.txt 2>&1
rm -rf $OUT/db_$X
echo "experiment $X"; cat $OUT/draw_overlap_experiment_$X.txt; grep -o '"value": [0-9.]*' $OUT/stats_$X.log | head -1
done
