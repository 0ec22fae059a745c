#!/bin/bash
# r12u: the closed loop's kernel durations as distributions (scripts/queue_gaps.py): how much of the step launch's 18.4 us mean are the ticks in which an env resets
set -u
TAG=${1:-r12u}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/db -o run -- python $R/scripts/probe_closed_loop.py 1024 3000 > $OUT/closed_loop_traced.log 2>&1
python $R/scripts/queue_gaps.py $OUT/db/run_results.db > $OUT/queue_gaps_closed_loop.txt 2>&1; rm -rf $OUT/db; cat $OUT/queue_gaps_closed_loop.txt
