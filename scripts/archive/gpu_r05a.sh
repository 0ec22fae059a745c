#!/bin/bash
# r05a: cost bins from the last pass's classification (feedback) with and without graded split
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "fb0 MV_RASTER_COST_FEEDBACK=0" "fb1 X=1" "fb1_g8 MV_RASTER_GRADED=1 MV_RASTER_GRADED_DIV=8" "fb1_g4 MV_RASTER_GRADED=1 MV_RASTER_GRADED_DIV=4" "fb1_g3 MV_RASTER_GRADED=1 MV_RASTER_GRADED_DIV=3" "fb0_g4 MV_RASTER_COST_FEEDBACK=0 MV_RASTER_GRADED=1 MV_RASTER_GRADED_DIV=4"; do
  set -- $cfg; N=$1; shift
  env "$@" MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_$N.json 2> $OUT/rt_$N.err
  echo "== $N"; grep "raster timing (\|decile" $OUT/rt_$N.err | cut -c1-300
  (cd /tmp; env "$@" MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/u.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/ks_$N.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
  grep -h "raster_fast\|step_kernel" $OUT/ks_$N.csv | cut -d, -f3-8
done
