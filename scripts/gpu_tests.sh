#!/bin/bash
# run a set of GPU tests: gpu_tests.sh <outdir> <pytest args...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest "$@" -m gpu -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -40 $OUT/pytest.log
