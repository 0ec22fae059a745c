#!/bin/bash
# r10v: r10p's best first-call schedule for the 20-step form (1,3,6,10) against the default (2,4,6), five runs each, interleaved
set -u
TAG=${1:-r10v}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for i in 1 2 3 4 5; do
  for S in 2,4,6 1,3,6,10 1,3,6 1,2,4,6,7; do
    MV_BENCH_CALL_SCHEDULE=$S timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 0 > $OUT/driver_${S}_$i.json 2> /dev/null
    python -c "
import json; d=json.loads(open('$OUT/driver_${S}_$i.json').read().strip().splitlines()[-1]); print('schedule $S run $i: %.2f M' % (d['value']/1e6))"
  done
done
