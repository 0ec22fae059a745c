#!/bin/bash
# r08t: a call of 16 ticks as TWO step launches of 8 (views by value: no write_views kernel in front of the launch; r08r's Empty timeline showed those tiny
# kernels taking 10-44 us beside the observation launch that had just started, and the step launch behind them arriving late) against one launch of 16
# (MV_X_TICKS_CHUNK=16); and 8 against 16 ticks per call again now that the simulation queue carries no markers.  High-priority simulation stream throughout.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08t; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_pipelining_gpu.py tests/test_parity_gpu.py tests/test_refill_protocol_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
run() { local tag=$1; shift
  for s in TowerBuilding ObstaclesHard ObstaclesEasy Empty Rearrange Collect Sokoban HexMemory; do
    $B --scenario $s "$@" > $OUT/${tag}_${s}_bench.json 2> /dev/null
  done
}
run c8_b16 --batch 16
MV_X_TICKS_CHUNK=16 run c16_b16 --batch 16
run b8 --batch 8
$B --steps 20 --warmup 5 > $OUT/c8_driver_style_1_bench.json 2> /dev/null
$B --steps 20 --warmup 5 > $OUT/c8_driver_style_2_bench.json 2> /dev/null
MV_X_TICKS_CHUNK=16 $B --steps 20 --warmup 5 > $OUT/c16_driver_style_1_bench.json 2> /dev/null
MV_X_TICKS_CHUNK=16 $B --steps 20 --warmup 5 > $OUT/c16_driver_style_2_bench.json 2> /dev/null
for f in $OUT/*_bench.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', d['steps'], d['config'].get('ticks_per_call'))
except Exception as e: print('$f', 'failed', e)
"; done
