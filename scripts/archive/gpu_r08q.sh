#!/bin/bash
# r08q: (a) episode uploads of consecutive envs in one strided copy (refill_episodes): refill / parity tests, Empty over 700 and 2000 steps (the mass reset at
# tick 900 falls into the second only); (b) one copy stream per group: Mixed 64 x 64, timeline; (c) the ticks-per-call rule on the 512-env lines and Sokoban
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08q; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_refill_protocol_gpu.py tests/test_multitask_gpu.py tests/test_empty_parity_gpu.py tests/test_obstacles_parity_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
$B --scenario Empty --steps 700 --warmup 96 > $OUT/Empty_700_steps_bench.json 2> /dev/null
$B --scenario Empty > $OUT/Empty_bench.json 2> /dev/null
$B --scenario Empty --steps 4000 > $OUT/Empty_4000_steps_bench.json 2> /dev/null
$B --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> /dev/null
$B --scenario Mixed --obs 64 64 --steps 240 --warmup 48 > $OUT/mixed_64_240_steps_bench.json 2> /dev/null
$B --scenario Mixed4 --obs 64 64 > $OUT/mixed4_64_bench.json 2> /dev/null
$B --scenario Mixed --obs 128 128 > $OUT/mixed_128_bench.json 2> /dev/null
$B --scenario Sokoban > $OUT/Sokoban_bench.json 2> /dev/null
$B --envs-per-gpu 512 > $OUT/tower_512_bench.json 2> /dev/null
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/obstacles_hard_512_bench.json 2> /dev/null
$B --scenario ObstaclesEasy > $OUT/ObstaclesEasy_bench.json 2> /dev/null
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $OUT/db_s -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 240 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/mixed.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_s/run_results.db 80 60 > $OUT/timeline_mixed_64.txt 2>> $OUT/mixed.log
 python $R/scripts/rocpd_summary.py $OUT/db_s/run_results.db > $OUT/mixed_64_kernel_stats.csv 2>> $OUT/mixed.log; rm -rf $OUT/db_s)
for f in $OUT/*_bench.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', d['steps'], 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), d['config'].get('ticks_per_call'))
except Exception as e: print('$f', 'failed', e)
"; done
