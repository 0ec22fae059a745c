#!/bin/bash
# (the MV_GLIST_BATCH_SPLIT hook was added for this measurement and not kept: the rule -- two workgroups per long-list frame -- is the best of 1 / 2 / 4)
# r10o: workgroups per frame of the long-list one-launch passes (MV_GLIST_BATCH_SPLIT=1 / 2 / 4; the rule says 2)
set -u
TAG=${1:-r10o}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for S in 1 2 4; do
    MV_GLIST_BATCH_SPLIT=$S run collect_split${S}_$i $B --scenario Collect
    MV_GLIST_BATCH_SPLIT=$S run hexmemory_split${S}_$i $B --scenario HexMemory
  done
done
