#!/bin/bash
# r09j: overlapped passes (mv_set_pass_overlap) for TowerBuilding, in the steady state and in the driver's 20-step form (four short calls: four launch tails)
set -u
TAG=${1:-r09j}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 16"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', d['ms_per_step'])
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2 3; do
  run driver_off_$i $B --gpus 1 --steps 20 --warmup 5 --pass-overlap off
  run driver_on_$i $B --gpus 1 --steps 20 --warmup 5 --pass-overlap on
done
for i in 1 2; do
  run steady_off_$i $B --pass-overlap off
  run steady_on_$i $B --pass-overlap on
done
MV_BENCH_CALL_SCHEDULE=2,4,6 run driver_on_b8 $B --gpus 1 --steps 20 --warmup 5 --pass-overlap on --batch 8
run driver_off_b8 $B --gpus 1 --steps 20 --warmup 5 --pass-overlap off --batch 8
MV_STEP_PIPE=1 run driver_on_pipe $B --gpus 1 --steps 20 --warmup 5 --pass-overlap on
run steps100_off $B --gpus 1 --steps 100 --warmup 10 --pass-overlap off
run steps100_on $B --gpus 1 --steps 100 --warmup 10 --pass-overlap on
