#!/bin/bash
# r11a: the default bench line's transparency legs with the overlapped passes switched off (and their two streams destroyed) before them, against --pass-overlap off
# throughout (r11z: with the pass streams left alive the legs that follow ran slower: env_step_device 7.5 M against 16.7 M)
set -u
TAG=${1:-r11a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_pipelining_gpu.py -m gpu -q -x -k overlapped > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
show() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1'.split('/')[-1], round(d['value']/1e6,2), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})"; }
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline > $OUT/tower_auto_$i.json 2> $OUT/tower_auto_$i.err; show $OUT/tower_auto_$i.json
  timeout 600 python bench.py --no-cpu-baseline --pass-overlap off > $OUT/tower_off_$i.json 2> $OUT/tower_off_$i.err; show $OUT/tower_off_$i.json
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_auto_$i.json 2> /dev/null; show $OUT/driver_auto_$i.json
done
