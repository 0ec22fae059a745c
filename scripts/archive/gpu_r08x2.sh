#!/bin/bash
# r08x2: one status read-back in flight against a re-issue every period (MV_X_REISSUE=1: as until round 5), each with three run-ahead bounds
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08x2; mkdir -p $OUT; cd $R
B="timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 --no-extra-legs"
for re in 0 1; do for cfg in "32 3" "64 6" "32 32"; do
  set -- $cfg
  export MV_X_BOUND_TICKS=$1 MV_X_BOUND_CALLS=$2
  if [ $re = 1 ]; then export MV_X_REISSUE=1; else unset MV_X_REISSUE; fi
  tag=re${re}_f$1_c$2
  for i in 1 2; do
    $B > $OUT/${tag}_tower_${i}_bench.json 2> /dev/null
    $B --scenario Empty > $OUT/${tag}_Empty_${i}_bench.json 2> /dev/null
  done
done; done
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2))
"; done
