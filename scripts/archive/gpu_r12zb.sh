#!/bin/bash
# which kernels a batched call launches with the pipelining off (MV_PIPELINE=0), and how long they are alone on the chip: kernel stats of a 256-step run
set -u
TAG=${1:-r12zb}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db -o run -- python $R/bench.py --steps 256 --warmup 32 --no-cpu-baseline --profile-steps 64 --no-extra-legs > $OUT/bench.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db/run_results.db > $OUT/kernel_stats_unpipelined_batched.csv 2>> $OUT/bench.log; rm -rf $OUT/db
cut -c1-150 $OUT/kernel_stats_unpipelined_batched.csv | head -12
tail -1 $OUT/bench.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['roofline']['avg_launch_ms'], d['roofline_physics']['avg_launch_ms'], d['config'])"
