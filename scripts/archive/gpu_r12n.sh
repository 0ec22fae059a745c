#!/bin/bash
# r12n: configs[4] on two host cores (taskset -c 0,1 = a 16-CPU quota / 8 ranks): Mixed / Mixed4 64 x 64 with the Collect members' episodes from the host feeder (MV_COLLECT_DEVICE_GEN=0) and from the device (the rule's choice there), three runs each; the same on all cores
set -u
TAG=${1:-r12n}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --obs 64 64"
for i in 1 2 3; do for S in Mixed Mixed4; do
  MV_COLLECT_DEVICE_GEN=0 taskset -c 0,1 $B --scenario $S > $OUT/${S}_host_2cores_$i.json 2> /dev/null
  taskset -c 0,1 $B --scenario $S > $OUT/${S}_rule_2cores_$i.json 2> /dev/null
done; done
for S in Mixed Mixed4; do
  MV_COLLECT_DEVICE_GEN=0 $B --scenario $S > $OUT/${S}_host_all.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=1 $B --scenario $S > $OUT/${S}_device_all.json 2> /dev/null
done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M')
except Exception as e: print('$f', 'failed', e)
"; done
