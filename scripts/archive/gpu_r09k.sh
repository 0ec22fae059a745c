#!/bin/bash
# r09k: the ticks-per-call rule (mv_recommended_ticks_per_call) against the lighter observation pass: 8 against 16 ticks per call where the rule decides
set -u
TAG=${1:-r09k}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 32"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', d['config'].get('ticks_per_call'), d['config'].get('overlapped_passes'))
except Exception as e: print('$name', 'failed', e)
PY
}
for b in 8 16; do
  run tower512_b$b $B --envs-per-gpu 512 --batch $b
  run tower1024_b$b $B --batch $b
  run tower2048_b$b $B --envs-per-gpu 2048 --batch $b
  run tower4096_b$b $B --envs-per-gpu 4096 --batch $b
  run tower512x4_b$b $B --envs-per-gpu 512 --agents 4 --batch $b
  run empty_b$b $B --scenario Empty --batch $b
  run sokoban_b$b $B --scenario Sokoban --batch $b
  run obsthard1024_b$b $B --scenario ObstaclesHard --batch $b
  run obsthard512_b$b $B --scenario ObstaclesHard --envs-per-gpu 512 --batch $b
  run obsthard512_noov_b$b $B --scenario ObstaclesHard --envs-per-gpu 512 --batch $b --pass-overlap off
  run rearrange_b$b $B --scenario Rearrange --batch $b
  run collect_b$b $B --scenario Collect --batch $b
  run hexmemory_b$b $B --scenario HexMemory --batch $b
done
run sokoban_b8_noov $B --scenario Sokoban --batch 8 --pass-overlap off
run sokoban_b16_noov $B --scenario Sokoban --batch 16 --pass-overlap off
run obsthard1024_b16_noov $B --scenario ObstaclesHard --batch 16 --pass-overlap off
