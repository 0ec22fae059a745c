#!/bin/bash
# r09e: r09d's runs with the adjacency test (a frame cut into pieces deals its tiles out in fours: the previous lane's tile is not always the left neighbour)
set -u
TAG=${1:-r09e}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests/test_fast_pixels_gpu.py tests/test_canonical_frames_gpu.py tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_multitask_gpu.py tests/test_rearrange_parity_gpu.py tests/test_sokoban_parity_gpu.py tests/test_empty_parity_gpu.py tests/test_obstacles_parity_gpu.py tests/test_pipelining_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
run tower_1 $B
run tower_2 $B
run empty $B --scenario Empty
run tower512 $B --envs-per-gpu 512
run tower4096 $B --envs-per-gpu 4096
run tower512x4 $B --envs-per-gpu 512 --agents 4
run obsthard $B --scenario ObstaclesHard
run obsthard512 $B --scenario ObstaclesHard --envs-per-gpu 512
run rearrange $B --scenario Rearrange
run rearrange_2 $B --scenario Rearrange
run sokoban $B --scenario Sokoban
run driver timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 16
run driver_2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 16
# one rank's share of a 16-CPU host with 8 ranks: two cores for the Python thread and the episode feeders
python scripts/probe_generators.py 400 > $OUT/probe_generators.txt 2>&1
taskset -c 0 python scripts/probe_generators.py 400 > $OUT/probe_generators_1core.txt 2>&1
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
for s in Collect ObstaclesHard HexMemory Rearrange Sokoban; do
  run ${s}_all_cores $B --scenario $s
  run ${s}_2_cores taskset -c 0,1 $B --scenario $s
done
run mixed64_all_cores $B --scenario Mixed --obs 64 64
run mixed64_2_cores taskset -c 0,1 $B --scenario Mixed --obs 64 64
cat $OUT/probe_generators.txt
