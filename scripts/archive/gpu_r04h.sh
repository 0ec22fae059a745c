#!/bin/bash
set -u
TAG=${1:-r04h}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for NS in 0 1; do for S in 2 4; do
  MV_RASTER_NOSORT=$NS MV_RASTER_SPLIT=$S MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_ns${NS}_s$S.json 2> $OUT/rt_ns${NS}_s$S.err
  echo "nosort $NS split $S:"; grep "raster timing (" $OUT/rt_ns${NS}_s$S.err
  (cd /tmp; MV_RASTER_NOSORT=$NS MV_RASTER_SPLIT=$S MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/u.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/tower_unpipelined_ns${NS}_s${S}_kernel_stats.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
  grep -h raster_fast $OUT/tower_unpipelined_ns${NS}_s${S}_kernel_stats.csv | cut -d, -f3-8
done; done
