#!/bin/bash
# r04i: the k observation passes of a batched call as one launch
set -u
TAG=${1:-r04i}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_pipelining_gpu.py tests/test_fast_pixels_gpu.py -x -q -k "not within_tolerance" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for BT in 1 0; do
  MV_RASTER_BATCH=$BT timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > $OUT/tower_batch$BT.json 2> $OUT/tower_batch$BT.err
  MV_RASTER_BATCH=$BT timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --gpus 1 --steps 20 --warmup 5 > $OUT/tower_batch${BT}_driver_style.json 2>> $OUT/tower_batch$BT.err
  MV_RASTER_BATCH=$BT timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --agents 4 --envs-per-gpu 512 > $OUT/tower_512x4_batch$BT.json 2>> $OUT/tower_batch$BT.err
  MV_RASTER_BATCH=$BT timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_batch$BT.json 2>> $OUT/tower_batch$BT.err
done
for f in $OUT/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), (d.get('roofline') or {}).get('avg_launch_ms'), d['config']['ticks_per_call'])" 2>/dev/null)"; done
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/u.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/tower_pipelined_kernel_stats.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
cat $OUT/tower_pipelined_kernel_stats.csv | cut -c1-160 | head -8
