#!/bin/bash
# r12q: two host cores, ONE feeder thread (r12p: ObstaclesHard 1024 envs 29.4 M obs/s with one, 21.5 - 25.6 with the rule's two): Collect host-fed / device-fed, Mixed, Mixed4, three runs each
set -u
TAG=${1:-r12q}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2 3; do
  MV_COLLECT_DEVICE_GEN=0 MV_FEEDER_THREADS=1 taskset -c 0,1 $B --scenario Collect > $OUT/collect_host_threads1_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=0 MV_FEEDER_THREADS=2 taskset -c 0,1 $B --scenario Collect > $OUT/collect_host_threads2_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=1 taskset -c 0,1 $B --scenario Collect > $OUT/collect_device_$i.json 2> /dev/null
  for S in Mixed Mixed4; do for T in 1 2; do
    MV_COLLECT_DEVICE_GEN=0 MV_FEEDER_THREADS=$T taskset -c 0,1 $B --scenario $S --obs 64 64 > $OUT/${S}_host_threads${T}_$i.json 2> /dev/null
  done; MV_COLLECT_DEVICE_GEN=1 MV_FEEDER_THREADS=1 taskset -c 0,1 $B --scenario $S --obs 64 64 > $OUT/${S}_device_threads1_$i.json 2> /dev/null
  done
done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M')
except Exception as e: print('$f', 'failed', e)
"; done
