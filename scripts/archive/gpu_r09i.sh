#!/bin/bash
# r09i: the wave-local frame setup's rounds without the per-round s_waitcnt vmcnt(0) (sync_rounds), the hex tick's cluster pre-test
set -u
TAG=${1:-r09i}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 2400 python -m pytest tests/test_hex_parity_gpu.py tests/test_collect_parity_gpu.py tests/test_obstacles_parity_gpu.py tests/test_parity_gpu.py tests/test_pipelining_gpu.py tests/test_soak_gpu.py tests/test_multitask_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for s in HexMemory HexExplore Collect; do run $s $B --scenario $s; MV_PIPELINE=0 run ${s}_alone $B --scenario $s --steps 400; done
run tower $B
MV_PIPELINE=0 run tower_alone $B --steps 400
run empty $B --scenario Empty
MV_PIPELINE=0 run empty_alone $B --scenario Empty --steps 400
run obsthard $B --scenario ObstaclesHard
run obsthard512 $B --scenario ObstaclesHard --envs-per-gpu 512
run sokoban $B --scenario Sokoban
run mixed64_1 $B --scenario Mixed --obs 64 64
run mixed64_2 $B --scenario Mixed --obs 64 64
run mixed4_64 $B --scenario Mixed4 --obs 64 64
