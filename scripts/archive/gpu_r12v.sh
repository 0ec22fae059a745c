#!/bin/bash
# r12v: TowerBuilding's swap-in without forming the chunk's empty layers cell by cell: the closed loop (probe x 3, trace + scripts/queue_gaps.py), the bench's legs, the tower parity tests
set -u
TAG=${1:-r12v}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for i in 1 2 3; do timeout 300 python $R/scripts/probe_closed_loop.py 1024 3000 2>&1 | tail -1; done
timeout 300 rocprofv3 --kernel-trace -d $OUT/db -o run -- python $R/scripts/probe_closed_loop.py 1024 3000 > $OUT/closed_loop_traced.log 2>&1
python $R/scripts/queue_gaps.py $OUT/db/run_results.db > $OUT/queue_gaps_closed_loop.txt 2>&1; rm -rf $OUT/db; grep "step_kernel\|busy" $OUT/queue_gaps_closed_loop.txt
cd $R
timeout 600 python bench.py --no-cpu-baseline --profile-steps 128 > $OUT/tower_bench_legs.json 2> /dev/null
python -c "import json; d=json.load(open('$OUT/tower_bench_legs.json')); print(round(d['value']/1e6,2), 'M', {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})"
timeout 2400 python -m pytest tests -m gpu -q -x -k "tower or Tower or reset or full_size or canonical or refill or soak or pipelin or surface" > $OUT/pytest_subset.log 2>&1; tail -3 $OUT/pytest_subset.log
