#!/bin/bash
# r12l: the device generator's tests (incl. the group), then every GPU test that touches Collect / the refill protocol / groups with the device generator FORCED on (MV_COLLECT_DEVICE_GEN=1) and the soak
set -u
TAG=${1:-r12l}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_collect_draw_gpu.py -q -x > $OUT/pytest_draw.log 2>&1; echo "rc=$?" >> $OUT/pytest_draw.log; tail -4 $OUT/pytest_draw.log
MV_COLLECT_DEVICE_GEN=1 timeout 2400 python -m pytest tests/test_collect_parity_gpu.py tests/test_refill_protocol_gpu.py tests/test_multitask_gpu.py tests/test_full_size_oracle_gpu.py tests/test_capacity_flags_gpu.py tests/test_soak_gpu.py tests/test_py_surface_gpu.py -m gpu -q -k "ollect or ulti or ixed or group or starv or short or soak or capacity" > $OUT/pytest_forced_device_gen.log 2>&1; echo "rc=$?" >> $OUT/pytest_forced_device_gen.log; tail -6 $OUT/pytest_forced_device_gen.log
