#!/bin/bash
# r12w: the double-buffered closed loop reads 11.9 M obs/s behind a headline with overlapped passes and 14.2 M behind one without: which hardware queues the halves run on (scripts/queues_of_halves.py), 600-step runs
set -u
TAG=${1:-r12w}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for M in on off; do
timeout 600 rocprofv3 --kernel-trace -d $OUT/db_$M -o run -- python $R/bench.py --steps 600 --warmup 64 --no-cpu-baseline --profile-steps 0 --pass-overlap $M > $OUT/bench_$M.log 2>&1
python $R/scripts/queues_of_halves.py $OUT/db_$M/run_results.db > $OUT/queues_of_halves_overlap_$M.txt 2>&1; rm -rf $OUT/db_$M
echo "pass overlap $M"; cat $OUT/queues_of_halves_overlap_$M.txt; grep -o '"value_closed_loop[a-z_]*": [0-9.]*' $OUT/bench_$M.log
done
