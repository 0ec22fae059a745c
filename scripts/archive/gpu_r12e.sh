#!/bin/bash
# r12e: the device generator's remaining 5 % against the host feeder on all cores (r12d: 16.0 / 16.8 M): hardware queues 8 / 12 / 16 / 24, passes not overlapped
set -u
TAG=${1:-r12e}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario Collect"
for Q in 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$Q MV_COLLECT_DEVICE_GEN=1 $B > $OUT/collect_device_q$Q.json 2> /dev/null
  GPU_MAX_HW_QUEUES=$Q MV_COLLECT_DEVICE_GEN=0 $B > $OUT/collect_host_q$Q.json 2> /dev/null
done
MV_COLLECT_DEVICE_GEN=1 $B --pass-overlap off > $OUT/collect_device_no_overlap.json 2> /dev/null
MV_COLLECT_DEVICE_GEN=0 $B --pass-overlap off > $OUT/collect_host_no_overlap.json 2> /dev/null
for f in $OUT/collect_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
