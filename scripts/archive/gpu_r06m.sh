#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd $R
for e in "X=1" "MV_ATTACH_DONE=0" "MV_STEP_TICKS=0"; do
  echo "== $e"; env $e timeout 120 python scripts/probe_host_calls.py 1024 40 2>&1 | grep -v amdgpu.ids
done
echo "== 512"; timeout 120 python scripts/probe_host_calls.py 512 40 2>&1 | grep -v amdgpu.ids
