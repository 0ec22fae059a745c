#!/bin/bash
# r13b: with the wave-parallel slab merge: every GPU test that touches Collect / the refill protocol / groups / the soak with the device generator forced on; Collect and Mixed4 on two cores
set -u
TAG=${1:-r13b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
MV_COLLECT_DEVICE_GEN=1 timeout 2400 python -m pytest tests/test_collect_draw_gpu.py tests/test_collect_parity_gpu.py tests/test_refill_protocol_gpu.py tests/test_multitask_gpu.py tests/test_full_size_oracle_gpu.py tests/test_capacity_flags_gpu.py tests/test_soak_gpu.py tests/test_py_surface_gpu.py -m gpu -q > $OUT/pytest_forced_device_gen.log 2>&1; echo "rc=$?" >> $OUT/pytest_forced_device_gen.log; tail -4 $OUT/pytest_forced_device_gen.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for i in 1 2 3; do
  taskset -c 0,1 $B --scenario Collect > $OUT/collect_2cores_$i.json 2> /dev/null
  taskset -c 0,1 $B --scenario Mixed4 --obs 64 64 > $OUT/mixed4_2cores_$i.json 2> /dev/null
done
for f in $OUT/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), 'M', d['config'].get('host_generator_threads'))
except Exception as e: print('$f', 'failed', e)
"; done
