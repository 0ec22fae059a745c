#!/bin/bash
# r10za: overlapped passes (mv_set_pass_overlap) for the scenarios the rule leaves out -- Collect, HexMemory, HexExplore, Rearrange, TowerBuilding -- on / off
set -u
TAG=${1:-r10za}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), d['config'].get('overlapped_passes'))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for S in Collect HexMemory HexExplore Rearrange; do
    for O in off on; do
      run ${S}_overlap_${O}_$i $B --scenario $S --pass-overlap $O
    done
  done
done
