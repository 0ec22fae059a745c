#!/bin/bash
# r12f: collect_draw_kernel at wave priority 3 (r12d / r12e: the device-fed Collect gym 16.0 M obs/s, the host-fed 16.8 M on all cores, 13.0 M on two)
set -u
TAG=${1:-r12f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario Collect"
for i in 1 2 3; do
  MV_COLLECT_DEVICE_GEN=1 $B > $OUT/collect_device_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=0 $B > $OUT/collect_host_$i.json 2> /dev/null
done
MV_COLLECT_DEVICE_GEN=1 taskset -c 0,1 $B > $OUT/collect_device_2cores.json 2> /dev/null
for f in $OUT/collect_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
