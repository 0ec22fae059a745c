#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04x; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t -o run -- python $R/scripts/probe_closed_loop.py 1024 100 > $OUT/closed.log 2>&1
python $R/scripts/kernel_timeline.py $OUT/db_t/run_results.db 30 30 > $OUT/timeline_closed_loop.txt 2>> $OUT/closed.log
rm -rf $OUT/db_t
tail -2 $OUT/closed.log | head -1; cat $OUT/timeline_closed_loop.txt | cut -c1-110
