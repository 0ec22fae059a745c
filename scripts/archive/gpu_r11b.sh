#!/bin/bash
# r11b: the legs that make NEW gyms after a gym that used overlapped passes (two more streams in the process) with 8 / 16 hardware queues instead of HIP's 4; and the
# spread of the driver's 20-step form with overlapped passes on / off, six runs each
set -u
TAG=${1:-r11b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
show() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], round(d['value']/1e6,2), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})"; }
for Q in 8 16; do
  GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --no-cpu-baseline > $OUT/tower_auto_q$Q.json 2> $OUT/tower_auto_q$Q.err; show $OUT/tower_auto_q$Q.json
done
for i in 1 2 3 4 5 6; do
  for O in on off; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 16 --pass-overlap $O > $OUT/driver_${O}_$i.json 2> /dev/null; show $OUT/driver_${O}_$i.json
  done
done
