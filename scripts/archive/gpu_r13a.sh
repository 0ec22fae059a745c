#!/bin/bash
# r13a: collect_draw_kernel's slab merge by all 64 lanes (draw_slabs_wave): the kernel and gym tests, the phases (MV_DRAW_TIMING), Collect 1024 envs device- / host-fed, linger 0 / 15
set -u
TAG=${1:-r13a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_collect_draw_gpu.py -q -x -s > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
grep -h "collect_draw_kernel\|passed\|failed\|rc=\|Error" $OUT/pytest.log | tail -8
MV_DRAW_TIMING=1 timeout 600 python -m pytest tests/test_collect_draw_gpu.py -q -s -x -k "full_batch" 2>&1 | grep "draw timing"
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario Collect"
for i in 1 2 3; do
  MV_COLLECT_DEVICE_GEN=1 $B > $OUT/collect_device_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=1 MV_DRAW_LINGER_MS=0 $B > $OUT/collect_device_linger0_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=1 MV_DRAW_LINGER_MS=5 $B > $OUT/collect_device_linger5_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=0 $B > $OUT/collect_host_$i.json 2> /dev/null
done
for f in $OUT/collect_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
