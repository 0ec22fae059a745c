#!/bin/bash
# r12p: ObstaclesHard on two host cores with one and with two feeder threads (the rule gives cores / ranks = 2), four runs each
set -u
TAG=${1:-r12p}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario ObstaclesHard"
for i in 1 2 3 4; do for N in 512 1024; do for T in 1 2; do
  MV_FEEDER_THREADS=$T taskset -c 0,1 $B --envs-per-gpu $N > $OUT/oh_${N}_threads${T}_$i.json 2> /dev/null
done; done; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M')
except Exception as e: print('$f', 'failed', e)
"; done
