#!/bin/bash
# r11d: Mixed4 64 x 64 again (r10z: 21.8 M obs/s, r11z: 19.1 twice): four runs each with 4 and 8 hardware queues
set -u
TAG=${1:-r11d}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2 3 4; do
  for Q in 4 8; do
    GPU_MAX_HW_QUEUES=$Q $B --scenario Mixed4 --obs 64 64 > $OUT/mixed4_q${Q}_$i.json 2> /dev/null
    python -c "
import json; d=json.loads(open('$OUT/mixed4_q${Q}_$i.json').read().strip().splitlines()[-1]); print('mixed4 queues $Q run $i: %.2f M' % (d['value']/1e6))"
  done
done
