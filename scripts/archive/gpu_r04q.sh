#!/bin/bash
# r04q: the whole GPU suite (no -x: list every failure)
set -u
TAG=${1:-r04q}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -30 $OUT/pytest_gpu.log
