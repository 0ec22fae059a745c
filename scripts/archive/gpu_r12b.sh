#!/bin/bash
# r12b: as r12a with the device-drawn episodes copied into the ring on the simulation stream (r12a: on the copy stream: 13.7 M against the host feeder's 16.8 M)
set -u
TAG=${1:-r12b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_collect_draw_gpu.py tests/test_collect_draw.py -q -s -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
grep -h "collect_draw_kernel\|passed\|failed\|rc=\|Error\|assert" $OUT/pytest.log | tail -20
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario Collect"
for i in 1 2; do
  MV_COLLECT_DEVICE_GEN=0 $B > $OUT/collect_host_all_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=1 $B > $OUT/collect_device_all_$i.json 2> $OUT/collect_device_all_$i.err
  MV_COLLECT_DEVICE_GEN=0 taskset -c 0,1 $B > $OUT/collect_host_2cores_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=1 taskset -c 0,1 $B > $OUT/collect_device_2cores_$i.json 2> /dev/null
done
taskset -c 0,1 $B > $OUT/collect_auto_2cores.json 2> /dev/null
for f in $OUT/collect_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
tail -3 $OUT/collect_device_all_1.err
