#!/bin/bash
# r10m: the union step launch with the long-list gyms' envs first (their workgroups are the launch's longest; MV_UNION_LONG_FIRST=0: in the group's order)
set -u
TAG=${1:-r10m}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_multitask_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -2 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario Mixed"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2 3 4; do
  MV_UNION_LONG_FIRST=2 run mixed64_long_last_$i $B --obs 64 64
  MV_UNION_LONG_FIRST=0 run mixed64_group_order_$i $B --obs 64 64
done
for i in 1 2; do
  MV_UNION_LONG_FIRST=2 run mixed128_long_last_$i $B
  MV_UNION_LONG_FIRST=0 run mixed128_group_order_$i $B
done
