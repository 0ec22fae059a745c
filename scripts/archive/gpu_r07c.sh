#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07c; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for sched in "1,3" "1,2,4" "1,2,4,6" "2,4,6" "1,2,3,6" "2,2,4,4" "1,1,2,4,4" "1,2,2,3,4" "3,8" "1,3,4,4"; do
  vals=""
  for rep in 1 2 3; do
    v=$(MV_BENCH_CALL_SCHEDULE=$sched timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 0 2>/dev/null | python -c "import json,sys; print('%.2f'%(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']/1e6))")
    vals="$vals $v"
  done
  echo "schedule $sched:$vals"
done
