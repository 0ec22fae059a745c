#!/bin/bash
# light re-measure of the headline after the last code changes of round 2 (exact mode pipelined, status polling): gpu_profile_r02c.sh <tag>
set -u
TAG=${1:-r02c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
cd $R; timeout 400 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_tower_stats -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 > $OUT/tower_stats.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db_tower_stats/run_results.db > $OUT/tower_kernel_stats.csv 2>> $OUT/tower_stats.log
cd $R
timeout 300 python bench.py --pixels exact --no-cpu-baseline > $OUT/tower_exact_bench.json 2> $OUT/tower_exact_bench.err
timeout 300 python bench.py --agents 4 --envs-per-gpu 512 --no-cpu-baseline > $OUT/tower_512x4_bench.json 2> $OUT/e1.err
timeout 300 python bench.py --scenario ObstaclesHard --envs-per-gpu 512 --no-cpu-baseline > $OUT/obstacles_hard_512_bench.json 2> $OUT/e2.err
timeout 300 python bench.py --scenario Mixed --obs 64 64 --no-cpu-baseline > $OUT/mixed_64_bench.json 2> $OUT/e3.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_driver_style_bench.json 2> $OUT/e4.err
rm -rf $OUT/db_*
for f in $OUT/*_bench.json; do echo $f; tail -1 $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,3),'M obs/s', round(d['ms_per_step'],4),'ms', 'unpipelined', d.get('value_unpipelined'), 'raster', round(d.get('roofline',{}).get('avg_launch_ms',0),4), 'step', round(d.get('roofline_physics',{}).get('avg_launch_ms',0),4))"; done
head -5 $OUT/tower_kernel_stats.csv
