#!/bin/bash
# r08r: who gets the chip first when a call's observation launch and the next call's step launch become ready together?  (r08q timeline: the step launch of a
# group call started 20-60 us behind the observation launch and then waited for its workgroups to drain: 550 us instead of 270.)  One event behind a call's
# step launches (no marker packets in front of the next step launch); MV_X_SIM_PRIORITY=high: a high-priority simulation stream; MV_X_COPY=own: a copy
# stream per group member as before.  Step-bound configurations: Mixed 64 x 64, Empty, 512 envs, Sokoban, ObstaclesHard 512; and the headline.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08r; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_refill_protocol_gpu.py tests/test_multitask_gpu.py tests/test_pipelining_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local tag=$1; shift
  $B --scenario Mixed --obs 64 64 > $OUT/${tag}_mixed_64_bench.json 2> /dev/null
  $B --scenario Mixed4 --obs 64 64 > $OUT/${tag}_mixed4_64_bench.json 2> /dev/null
  $B --scenario Empty > $OUT/${tag}_Empty_bench.json 2> /dev/null
  $B --envs-per-gpu 512 > $OUT/${tag}_tower_512_bench.json 2> /dev/null
  $B --scenario Sokoban > $OUT/${tag}_Sokoban_bench.json 2> /dev/null
  $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/${tag}_obstacles_hard_512_bench.json 2> /dev/null
  $B > $OUT/${tag}_tower_bench.json 2> /dev/null
  $B --steps 20 --warmup 5 > $OUT/${tag}_tower_driver_style_bench.json 2> /dev/null
}
run e0
MV_X_SIM_PRIORITY=high run e1
MV_X_COPY=own $B --scenario Mixed --obs 64 64 > $OUT/e2_mixed_64_bench.json 2> /dev/null
MV_X_COPY=own $B --scenario Mixed4 --obs 64 64 > $OUT/e2_mixed4_64_bench.json 2> /dev/null
MV_X_COPY=own MV_X_SIM_PRIORITY=high $B --scenario Mixed --obs 64 64 > $OUT/e3_mixed_64_bench.json 2> /dev/null
(cd /tmp; MV_X_SIM_PRIORITY=high timeout 300 rocprofv3 --kernel-trace -d $OUT/db_s -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 240 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/mixed_e1.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_s/run_results.db 80 60 > $OUT/timeline_mixed_64_e1.txt 2>> $OUT/mixed_e1.log; rm -rf $OUT/db_s)
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $OUT/db_s -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 240 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/mixed_e0.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_s/run_results.db 80 60 > $OUT/timeline_mixed_64_e0.txt 2>> $OUT/mixed_e0.log; rm -rf $OUT/db_s)
(cd /tmp; MV_X_SIM_PRIORITY=high timeout 300 rocprofv3 --kernel-trace -d $OUT/db_s -o run -- python $R/bench.py --scenario Empty --steps 480 --warmup 96 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/empty_e1.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_s/run_results.db 60 40 > $OUT/timeline_empty_e1.txt 2>> $OUT/empty_e1.log; rm -rf $OUT/db_s)
for f in $OUT/*_bench.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', d['steps'], 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), d['config'].get('ticks_per_call'))
except Exception as e: print('$f', 'failed', e)
"; done
