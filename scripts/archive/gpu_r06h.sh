#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06h; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 120 rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -n -i -B2 -A12 "pc.sampl" $OUT/avail.txt | head -60
