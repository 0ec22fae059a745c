#!/bin/bash
# r10zc: overlapped passes for TowerBuilding by env count and frame size, and on the driver's 20-step form (r10zb: 1024 envs 32.5 -> 34.4 M obs/s; 512 x 4 agents 28.8 -> 24.5)
set -u
TAG=${1:-r10zc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), d['config'].get('ticks_per_call'), d['config'].get('overlapped_passes'))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for O in off on; do
    for E in 256 512 768 2048 4096; do run tower_${E}_overlap_${O}_$i $B --envs-per-gpu $E --pass-overlap $O; done
    run tower_1024_128x72_overlap_${O}_$i $B --obs 128 72 --pass-overlap $O
    run tower_1024_64x64_overlap_${O}_$i $B --obs 64 64 --pass-overlap $O
    run tower_1024_k8_overlap_${O}_$i $B --batch 8 --pass-overlap $O
    run tower_512x2_overlap_${O}_$i $B --envs-per-gpu 512 --agents 2 --pass-overlap $O
    run single_bit_overlap_${O}_$i $B --policy single-bit --pass-overlap $O
  done
done
for i in 1 2 3 4; do
  for O in off on; do
    run driver_overlap_${O}_$i timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 16 --pass-overlap $O
  done
done
