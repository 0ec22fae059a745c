#!/bin/bash
# r12s: the closed loop over 3000 ticks (scripts/probe_closed_loop.py) under a kernel trace: where the caller's queue idles (scripts/queue_gaps.py) -- the leg reads 67.9 us per tick, the kernels of a tick add up to 59
set -u
TAG=${1:-r12s}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 python $R/scripts/probe_closed_loop.py 1024 3000 > $OUT/closed_loop_plain.log 2>&1; tail -1 $OUT/closed_loop_plain.log
timeout 300 rocprofv3 --kernel-trace -d $OUT/db -o run -- python $R/scripts/probe_closed_loop.py 1024 3000 > $OUT/closed_loop_traced.log 2>&1; grep "closed loop" $OUT/closed_loop_traced.log
python $R/scripts/queue_gaps.py $OUT/db/run_results.db > $OUT/queue_gaps_closed_loop.txt 2>&1; rm -rf $OUT/db; cat $OUT/queue_gaps_closed_loop.txt
