#!/bin/bash
# r11c: bench.py with 8 hardware queues by default and overlapped passes only for regions of four calls and more: the driver's form six times, the default line
# twice, and 8 against 4 queues on the configurations with the most streams in play
set -u
TAG=${1:-r11c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
show() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], round(d['value']/1e6,2), d['config'].get('overlapped_passes'), d['config'].get('hip_hardware_queues'), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})"; }
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 16 > $OUT/driver_$i.json 2> /dev/null; show $OUT/driver_$i.json
done
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline > $OUT/tower_$i.json 2> $OUT/tower_$i.err; show $OUT/tower_$i.json
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_legs.json 2> /dev/null; show $OUT/driver_legs.json
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for Q in 4 8; do
  for S in "oh512 --scenario ObstaclesHard --envs-per-gpu 512" "collect --scenario Collect" "mixed64 --scenario Mixed --obs 64 64" "sokoban --scenario Sokoban" "tower512x4 --envs-per-gpu 512 --agents 4" "hexmemory --scenario HexMemory"; do
    set -- $S; N=$1; shift
    GPU_MAX_HW_QUEUES=$Q $B "$@" > $OUT/${N}_q$Q.json 2> /dev/null; show $OUT/${N}_q$Q.json
  done
done
