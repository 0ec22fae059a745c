#!/bin/bash
# r12t: the gym's internal events with a device-scope release (hipEventReleaseToDevice) against the default system scope (MV_EVENT_SCOPE=system): the closed loop (probe + gaps), the headline, the driver's form, ObstaclesHard 512, Collect -- and the parity suites that would see a stale read
set -u
TAG=${1:-r12t}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for S in device system; do
  for i in 1 2 3; do MV_EVENT_SCOPE=$S timeout 300 python $R/scripts/probe_closed_loop.py 1024 3000 2>&1 | tail -1 | sed "s/^/$S scope: /"; done
done
timeout 300 rocprofv3 --kernel-trace -d $OUT/db -o run -- python $R/scripts/probe_closed_loop.py 1024 3000 > $OUT/closed_loop_traced.log 2>&1
python $R/scripts/queue_gaps.py $OUT/db/run_results.db > $OUT/queue_gaps_closed_loop_device_scope.txt 2>&1; rm -rf $OUT/db; cat $OUT/queue_gaps_closed_loop_device_scope.txt
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for S in device system; do for i in 1 2; do
  MV_EVENT_SCOPE=$S $B > $OUT/tower_${S}_$i.json 2> /dev/null
  MV_EVENT_SCOPE=$S $B --gpus 1 --steps 20 --warmup 5 > $OUT/driver_${S}_$i.json 2> /dev/null
  MV_EVENT_SCOPE=$S $B --batch 1 > $OUT/tower_single_step_${S}_$i.json 2> /dev/null
  MV_EVENT_SCOPE=$S $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/oh512_${S}_$i.json 2> /dev/null
  MV_EVENT_SCOPE=$S $B --scenario Collect > $OUT/collect_${S}_$i.json 2> /dev/null
done; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M')
except Exception as e: print('$f', 'failed', e)
"; done
timeout 2400 python -m pytest tests -m gpu -q -x -k "pipelin or full_size or soak or refill or closed or policy or overlap or distributed" > $OUT/pytest_subset.log 2>&1; tail -3 $OUT/pytest_subset.log
