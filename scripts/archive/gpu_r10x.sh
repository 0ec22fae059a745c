#!/bin/bash
# r10x: where does a TowerBuilding step launch's time go today (-DMV_TICK_TIMING: cycles per phase of the tick and of the frame setup, printed by mv_close), alone
# on the chip (MV_PIPELINE=0) and beside the passes
set -u
TAG=${1:-r10x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
MV_TICK_TIMING=1 MV_TICK_TIMING_SKIP=50 MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_ticktime.so $B --steps 300 --warmup 50 > $OUT/tower_alone_timing.json 2> $OUT/tower_alone_timing.err
grep "mv tick timing" $OUT/tower_alone_timing.err | cut -c1-400
MV_TICK_TIMING=1 MV_TICK_TIMING_SKIP=50 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_ticktime.so $B --steps 300 --warmup 50 > $OUT/tower_pipelined_timing.json 2> $OUT/tower_pipelined_timing.err
grep "mv tick timing" $OUT/tower_pipelined_timing.err | cut -c1-400
