#!/bin/bash
# r13c: with the faster draw kernel, the eight-scenario Mixed batch (128 Collect envs) on two cores with its Collect member host-fed / device-fed, five runs each
set -u
TAG=${1:-r13c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --obs 64 64 --scenario Mixed"
for i in 1 2 3 4 5; do
  MV_COLLECT_DEVICE_GEN=0 taskset -c 0,1 $B > $OUT/mixed_host_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=1 taskset -c 0,1 $B > $OUT/mixed_device_$i.json 2> /dev/null
done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M')
except Exception as e: print('$f', 'failed', e)
"; done
