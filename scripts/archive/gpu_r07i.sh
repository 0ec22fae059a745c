#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r07i; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 600 python scripts/soak.py 12000 > $OUT/soak.log 2>&1; echo "rc=$?" >> $OUT/soak.log; grep -v amdgpu.ids $OUT/soak.log | tail -20 | cut -c1-250
