#!/bin/bash
# r08x: how far ahead of the status read-back may the host run?  max(floor ticks, calls x k): rates of the 16-tick configurations and the soak's batched part
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08x; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 --no-extra-legs"
for cfg in "32 2" "32 3" "64 4" "64 6" "128 8" "32 32"; do
  set -- $cfg
  export MV_X_BOUND_TICKS=$1 MV_X_BOUND_CALLS=$2
  tag=f$1_c$2
  $B > $OUT/${tag}_tower_bench.json 2> /dev/null
  $B --scenario Empty > $OUT/${tag}_Empty_bench.json 2> /dev/null
  $B --scenario ObstaclesHard > $OUT/${tag}_obstacles_hard_1024_bench.json 2> /dev/null
  $B --batch 8 > $OUT/${tag}_tower_b8_bench.json 2> /dev/null
done
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2))
"; done
export MV_X_BOUND_TICKS=64 MV_X_BOUND_CALLS=4
timeout 900 python -m pytest tests/test_refill_protocol_gpu.py -x -q -m gpu -k "far_ahead" 2>&1 | tail -3
export MV_X_BOUND_TICKS=32 MV_X_BOUND_CALLS=32
timeout 900 python -m pytest tests/test_refill_protocol_gpu.py -q -m gpu -k "far_ahead" 2>&1 | tail -5
