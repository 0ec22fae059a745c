#!/bin/bash
set -u
TAG=${1:-r04p}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests/test_canonical_frames_gpu.py tests/test_full_size_oracle_gpu.py "tests/test_fast_pixels_gpu.py::test_fast_hires_within_tolerance" -x -q --durations=8 > $OUT/pytest_new.log 2>&1; echo "rc=$?" >> $OUT/pytest_new.log
tail -25 $OUT/pytest_new.log
