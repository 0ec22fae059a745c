#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export MV_RASTER_COST_FEEDBACK=0
for cfg in "base X=1" "t4s4 MV_RASTER_TAIL_DIV=4 MV_RASTER_TAIL_SPLIT=4" "t8s8 MV_RASTER_TAIL_DIV=8 MV_RASTER_TAIL_SPLIT=8" "t8s4 MV_RASTER_TAIL_DIV=8 MV_RASTER_TAIL_SPLIT=4" "t16s8 MV_RASTER_TAIL_DIV=16 MV_RASTER_TAIL_SPLIT=8"; do
  set -- $cfg; N=$1; shift
  (cd /tmp; env "$@" MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/u.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/ks_$N.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
  echo "$N: $(grep -h "raster_fast" $OUT/ks_$N.csv | cut -d, -f3-8)"
done
