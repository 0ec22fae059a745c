#!/bin/bash
# r12c: where an episode's time goes in collect_draw_kernel (MV_DRAW_TIMING: clock reads at the phases' ends)
set -u
TAG=${1:-r12c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
MV_DRAW_TIMING=1 timeout 600 python -m pytest tests/test_collect_draw_gpu.py -q -s -x -k "kernel or full_batch" > $OUT/pytest_timing.log 2>&1
grep -h "draw timing\|collect_draw_kernel\|passed\|failed" $OUT/pytest_timing.log
