#!/bin/bash
# r13d: HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) on the launch-bound forms: the closed loop (probe), one mv_step per tick, the driver's 20-step form, the headline; three runs each way
set -u
TAG=${1:-r13d}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
for i in 1 2 3; do for K in 0 1; do
  HIP_FORCE_DEV_KERNARG=$K timeout 300 python scripts/probe_closed_loop.py 1024 3000 2>&1 | tail -1 | sed "s/^/kernarg $K: /"
  HIP_FORCE_DEV_KERNARG=$K $B --batch 1 > $OUT/single_step_k${K}_$i.json 2> /dev/null
  HIP_FORCE_DEV_KERNARG=$K $B --gpus 1 --steps 20 --warmup 5 > $OUT/driver_k${K}_$i.json 2> /dev/null
  HIP_FORCE_DEV_KERNARG=$K $B > $OUT/tower_k${K}_$i.json 2> /dev/null
done; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M')
except Exception as e: print('$f', 'failed', e)
"; done
