#!/bin/bash
# r08k: timeline of batched group calls (Mixed 64 x 64): where does a call's time go?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08k; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
(cd /tmp; timeout 300 rocprofv3 --kernel-trace -d $OUT/db_s -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 240 --warmup 48 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/mixed.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_s/run_results.db 80 60 > $OUT/timeline_mixed_64.txt 2>> $OUT/mixed.log; rm -rf $OUT/db_s)
grep -v "copyBuffer\|fillBuffer" $OUT/timeline_mixed_64.txt | head -40
python scripts/probe_host_calls.py 2>/dev/null | tail -5
