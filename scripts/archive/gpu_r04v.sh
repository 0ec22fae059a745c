#!/bin/bash
# r04v: one eight-wave workgroup per frame (MV_RASTER_WIDE=1) against two four-wave halves (0)
set -u
TAG=${1:-r04v}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py tests/test_pipelining_gpu.py -x -q -k "planar or per_lane or full_size or step_n or ring or hires" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for WIDE in 1 0; do
  MV_RASTER_WIDE=$WIDE MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_wide$WIDE.json 2> $OUT/rt_wide$WIDE.err
  echo "wide $WIDE:"; grep "raster timing (" $OUT/rt_wide$WIDE.err | cut -c1-330
  (cd /tmp; MV_RASTER_WIDE=$WIDE MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/u.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/tower_unpipelined_wide${WIDE}_kernel_stats.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
  grep -h raster_fast $OUT/tower_unpipelined_wide${WIDE}_kernel_stats.csv | cut -d, -f3-8
  MV_RASTER_WIDE=$WIDE timeout 300 python bench.py --no-cpu-baseline > $OUT/tower_wide$WIDE.json 2> $OUT/tower_wide$WIDE.err
  python -c "import json; d=json.load(open('$OUT/tower_wide$WIDE.json')); print(round(d['value']/1e6,2), 'M', {k: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['avg_launch_ms'], d['roofline_physics']['avg_launch_ms'])"
done
