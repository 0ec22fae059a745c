#!/bin/bash
# r04g: tiles handed out dynamically within a workgroup
set -u
TAG=${1:-r04g}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py -x -q -k "planar or per_lane or full_size" > $OUT/pytest_planar.log 2>&1; echo "rc=$?" >> $OUT/pytest_planar.log
tail -3 $OUT/pytest_planar.log
for G in 0 1; do
  MV_RASTER_GRADED=$G MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_graded$G.json 2> $OUT/rt_graded$G.err
  echo "graded $G:"; grep "raster timing" $OUT/rt_graded$G.err
  (cd /tmp; MV_RASTER_GRADED=$G MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u$G -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_graded$G.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u$G/run_results.db > $OUT/tower_unpipelined_graded${G}_kernel_stats.csv 2>> $OUT/tower_unpipelined_graded$G.log; rm -rf $OUT/db_u$G)
  grep -h raster_fast $OUT/tower_unpipelined_graded${G}_kernel_stats.csv | cut -d, -f3-8
done
for S in 1 4; do
  (cd /tmp; MV_RASTER_SPLIT=$S MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_s$S -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/tower_unpipelined_split$S.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_s$S/run_results.db > $OUT/tower_unpipelined_split${S}_kernel_stats.csv 2>> $OUT/tower_unpipelined_split$S.log; rm -rf $OUT/db_s$S)
  echo "split $S: $(grep -h raster_fast $OUT/tower_unpipelined_split${S}_kernel_stats.csv | cut -d, -f3-8)"
done
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $OUT/tower.json 2> $OUT/tower.err
echo "pipelined: $(python -c "import json; d=json.load(open('$OUT/tower.json')); print(round(d['value']/1e6,2), d['roofline']['avg_launch_ms'], d['roofline_physics']['avg_launch_ms'])")"
