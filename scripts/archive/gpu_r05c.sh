#!/bin/bash
# r05c: fine-grained tail: the cheapest frames, last in the launch, in more pieces
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "base X=1" "t4s4 MV_RASTER_TAIL_DIV=4 MV_RASTER_TAIL_SPLIT=4" "t2s4 MV_RASTER_TAIL_DIV=2 MV_RASTER_TAIL_SPLIT=4" "t4s8 MV_RASTER_TAIL_DIV=4 MV_RASTER_TAIL_SPLIT=8" "t8s8 MV_RASTER_TAIL_DIV=8 MV_RASTER_TAIL_SPLIT=8" "t3s4 MV_RASTER_TAIL_DIV=3 MV_RASTER_TAIL_SPLIT=4"; do
  set -- $cfg; N=$1; shift
  env "$@" MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_$N.json 2> $OUT/rt_$N.err
  echo "== $N"; grep "raster timing (\|life by decile" $OUT/rt_$N.err | cut -c1-300
  (cd /tmp; env "$@" MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/u.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/ks_$N.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
  grep -h "raster_fast" $OUT/ks_$N.csv | cut -d, -f3-8
done
