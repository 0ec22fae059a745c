#!/bin/bash
# r10r: the Obstacles family's resident step kernel without its spills (-DMV_STEP_TICKS_WAVES_PER_SIMD=2: 252 VGPRs, no scratch; the product's 128-VGPR build
# spills 276 bytes per lane), alone and as the two-wave pipelined kernel (MV_STEP_PIPE=1: 194 VGPRs), at 512 envs (configs[2]'s share per GPU) and 1024
set -u
TAG=${1:-r10r}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
V=$R/megaverse_amd/_variants/libmv_obst2.so
for i in 1 2; do
  for E in 512 1024; do
    run oh${E}_base_$i $B --scenario ObstaclesHard --envs-per-gpu $E
    MV_LIB_PATH=$V run oh${E}_252vgpr_$i $B --scenario ObstaclesHard --envs-per-gpu $E
    MV_STEP_PIPE=1 MV_LIB_PATH=$V run oh${E}_252vgpr_pipe_$i $B --scenario ObstaclesHard --envs-per-gpu $E
    MV_STEP_PIPE=1 run oh${E}_pipe_$i $B --scenario ObstaclesHard --envs-per-gpu $E
  done
  run empty1024_base_$i $B --scenario Empty
  MV_LIB_PATH=$V run empty1024_252vgpr_$i $B --scenario Empty
  MV_STEP_PIPE=1 MV_LIB_PATH=$V run empty1024_252vgpr_pipe_$i $B --scenario Empty
done
