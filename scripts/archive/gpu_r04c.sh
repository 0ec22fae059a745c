#!/bin/bash
# r04c: split sweep with the classified raster (kernel alone, rocprofv3)
set -u
TAG=${1:-r04c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for S in 1 2 4 8; do
  (cd /tmp; MV_RASTER_SPLIT=$S MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u$S -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_split$S.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u$S/run_results.db > $OUT/tower_unpipelined_split${S}_kernel_stats.csv 2>> $OUT/tower_unpipelined_split$S.log; rm -rf $OUT/db_u$S)
  echo "split $S: $(grep -h raster_fast $OUT/tower_unpipelined_split${S}_kernel_stats.csv | cut -d, -f3-8)"
done
cd $R
for S in 2 4; do
  MV_RASTER_SPLIT=$S timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $OUT/tower_split$S.json 2> $OUT/tower_split$S.err
  echo "split $S pipelined: $(python -c "import json; d=json.load(open('$OUT/tower_split$S.json')); print(round(d['value']/1e6,2), d['roofline']['avg_launch_ms'], d['roofline_physics']['avg_launch_ms'])")"
done
