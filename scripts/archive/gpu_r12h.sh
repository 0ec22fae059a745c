#!/bin/bash
# r12h: a long kernel trace (4000 steps: every env resets two or three times) of the device-fed and the host-fed Collect gym: scripts/draw_overlap.py
set -u
TAG=${1:-r12h}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for M in 1 0; do
MV_COLLECT_DEVICE_GEN=$M timeout 300 rocprofv3 --kernel-trace -d $OUT/db_$M -o run -- python $R/bench.py --scenario Collect --steps 4000 --warmup 96 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/stats_$M.log 2>&1
python $R/scripts/draw_overlap.py $OUT/db_$M/run_results.db > $OUT/draw_overlap_devgen_$M.txt 2>&1
rm -rf $OUT/db_$M
cat $OUT/draw_overlap_devgen_$M.txt; tail -1 $OUT/stats_$M.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M')"
done
