#!/bin/bash
# r04f: graded split, the heaviest 1/d of the frames in four pieces
set -u
TAG=${1:-r04f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py -x -q -k "planar" > $OUT/pytest_planar.log 2>&1; echo "rc=$?" >> $OUT/pytest_planar.log
tail -3 $OUT/pytest_planar.log
for D in 16 8 4 3; do
  MV_RASTER_GRADED_DIV=$D MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_div$D.json 2> $OUT/rt_div$D.err
  echo "div $D:"; grep "raster timing" $OUT/rt_div$D.err
  (cd /tmp; MV_RASTER_GRADED_DIV=$D MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u$D -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_div$D.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u$D/run_results.db > $OUT/tower_unpipelined_div${D}_kernel_stats.csv 2>> $OUT/tower_unpipelined_div$D.log; rm -rf $OUT/db_u$D)
  grep -h raster_fast $OUT/tower_unpipelined_div${D}_kernel_stats.csv | cut -d, -f3-8
done
