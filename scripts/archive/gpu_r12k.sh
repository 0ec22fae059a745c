#!/bin/bash
# r12k: the device feeder lets a batch gather for 25 ms unless somebody waits (MV_DRAW_LINGER_MS: 10 / 50 as well): tests, Collect device- / host-fed, the long trace
set -u
TAG=${1:-r12k}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_collect_draw_gpu.py -q -s -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
grep -h "collect_draw_kernel\|passed\|failed\|rc=\|Error" $OUT/pytest.log | tail -8
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128 --scenario Collect"
for i in 1 2 3; do
  MV_COLLECT_DEVICE_GEN=1 $B > $OUT/collect_device_$i.json 2> /dev/null
  MV_COLLECT_DEVICE_GEN=0 $B > $OUT/collect_host_$i.json 2> /dev/null
done
MV_COLLECT_DEVICE_GEN=1 taskset -c 0,1 $B > $OUT/collect_device_2cores.json 2> /dev/null
for L in 0 10 50 100; do MV_DRAW_LINGER_MS=$L MV_COLLECT_DEVICE_GEN=1 $B > $OUT/collect_device_linger_$L.json 2> /dev/null; done
for f in $OUT/collect_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$f', 'failed', e)
"; done
cd /tmp
MV_COLLECT_DEVICE_GEN=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/db_1 -o run -- python $R/bench.py --scenario Collect --steps 4000 --warmup 96 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/stats_1.log 2>&1
python $R/scripts/draw_overlap.py $OUT/db_1/run_results.db > $OUT/draw_overlap_devgen_1.txt 2>&1
rm -rf $OUT/db_1
cat $OUT/draw_overlap_devgen_1.txt
