#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --gpus 1 --steps 20 --warmup 5"
run() { local N=$1; shift; echo "$N: $(env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))")"; }
for rep in 1 2; do
run "k2" $B
run "k4" $B --batch 4
run "k8" $B --batch 8
run "1,3,8" MV_BENCH_CALL_SCHEDULE=1,3,8 $B --batch 8
run "1,2,4,8" MV_BENCH_CALL_SCHEDULE=1,2,4,8 $B --batch 8
run "2,4,6" MV_BENCH_CALL_SCHEDULE=2,4,6 $B --batch 8
run "1,1,2,4" MV_BENCH_CALL_SCHEDULE=1,1,2,4 $B --batch 4
done
