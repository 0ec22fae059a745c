#!/bin/bash
# r10zb: overlapped passes for Empty and TowerBuilding (on / off), and the overlap test with the Hex scenarios and Sokoban added
set -u
TAG=${1:-r10zb}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_pipelining_gpu.py -m gpu -q -x -k overlapped > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), d['config'].get('overlapped_passes'))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2 3; do
  for O in off on; do
    run Empty_overlap_${O}_$i $B --scenario Empty --pass-overlap $O
    run TowerBuilding_overlap_${O}_$i $B --pass-overlap $O
    run Tower512x4_overlap_${O}_$i $B --envs-per-gpu 512 --agents 4 --pass-overlap $O
    run Collect72_overlap_${O}_$i $B --scenario Collect --obs 128 72 --pass-overlap $O
  done
done
