#!/bin/bash
# r10t: the whole GPU suite with the two-wave step kernels chosen by env count (step_pipe_enabled), and the bench lines of the configurations the rule changes
set -u
TAG=${1:-r10t}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
run obstacles_hard_512_bench $B --scenario ObstaclesHard --envs-per-gpu 512
run obstacles_hard_512_no_overlap_bench $B --scenario ObstaclesHard --envs-per-gpu 512 --pass-overlap off
run tower_512_bench $B --envs-per-gpu 512
run Empty_bench $B --scenario Empty
run tower_bench_short $B
run obstacles_hard_1024_bench $B --scenario ObstaclesHard
