#!/bin/bash
# r13e: the union step launch with the long-list gyms' envs software-pipelined (MV_UNION_PIPE=1: step_union_ticks_pipe_kernel, two waves per env) against the four-wave form: the group tests, Mixed / Mixed4 64 x 64 and Mixed 128 x 128, four runs each way
set -u
TAG=${1:-r13e}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
MV_UNION_PIPE=1 timeout 1800 python -m pytest tests/test_multitask_gpu.py tests/test_full_size_oracle_gpu.py tests/test_collect_draw_gpu.py -m gpu -q -k "ulti or ixed or group or union" > $OUT/pytest_union_pipe.log 2>&1; echo "rc=$?" >> $OUT/pytest_union_pipe.log; tail -4 $OUT/pytest_union_pipe.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2 3 4; do for P in 0 1; do
  MV_UNION_PIPE=$P $B --scenario Mixed --obs 64 64 > $OUT/mixed64_pipe${P}_$i.json 2> /dev/null
  MV_UNION_PIPE=$P $B --scenario Mixed4 --obs 64 64 > $OUT/mixed4_pipe${P}_$i.json 2> /dev/null
done; done
for P in 0 1; do MV_UNION_PIPE=$P $B --scenario Mixed --obs 128 128 > $OUT/mixed128_pipe${P}.json 2> /dev/null; done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
