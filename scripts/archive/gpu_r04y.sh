#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04y; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0 --agents 4 --envs-per-gpu 512"
run() { local N=$1; shift; env "$@" $B > $OUT/$N.json 2> $OUT/$N.err; echo "$N: $(python -c "import json; d=json.load(open('$OUT/$N.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))" 2>/dev/null)"; tail -2 $OUT/$N.err | head -1; }
run a4_base X=1
for Wv in 1 2 4; do run a4_multi_w$Wv MV_STEP_TICKS_MULTI=1 MV_STEP_TICKS_WAVES=$Wv; done
timeout 600 python -m pytest tests/test_pipelining_gpu.py -x -q 2>&1 | tail -2
MV_STEP_TICKS_MULTI=1 MV_STEP_TICKS_WAVES=2 timeout 600 python -m pytest tests/test_pipelining_gpu.py -x -q 2>&1 | tail -2
