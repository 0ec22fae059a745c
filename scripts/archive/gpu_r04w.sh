#!/bin/bash
set -u
TAG=${1:-r04w}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_pipelining_gpu.py tests/test_obstacles_parity_gpu.py tests/test_empty_parity_gpu.py tests/test_refill_protocol_gpu.py -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
run() { local N=$1; shift; env "$@" $B > $OUT/$N.json 2> $OUT/$N.err; echo "$N: $(python -c "import json; d=json.load(open('$OUT/$N.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))" 2>/dev/null)"; }
B="$B --scenario ObstaclesHard --envs-per-gpu 512"
run obst512_ticks1 X=1
run obst512_ticks0 MV_STEP_TICKS=0
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --scenario ObstaclesHard"
run obst1024_ticks1 X=1
run obst1024_ticks0 MV_STEP_TICKS=0
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --scenario Empty"
run empty_ticks1 X=1
run empty_ticks0 MV_STEP_TICKS=0
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --agents 4 --envs-per-gpu 512"
run a4 X=1
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --scenario Sokoban"
run sokoban X=1
