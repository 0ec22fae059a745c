#!/bin/bash
# r12x: the whole GPU suite with Collect's device generator FORCED on (MV_COLLECT_DEVICE_GEN=1), and the soak
set -u
TAG=${1:-r12x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
MV_COLLECT_DEVICE_GEN=1 timeout 3000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_forced_device_gen.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu_forced_device_gen.log
tail -4 $OUT/pytest_gpu_forced_device_gen.log
MV_COLLECT_DEVICE_GEN=1 timeout 1200 python scripts/soak.py 6000 > $OUT/soak_forced_device_gen.log 2>&1; tail -3 $OUT/soak_forced_device_gen.log
