#!/bin/bash
# r09l: runs of up to eight tiles (a whole tile row of a 128-pixel frame)
set -u
TAG=${1:-r09l}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1800 python -m pytest tests/test_fast_pixels_gpu.py tests/test_canonical_frames_gpu.py tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 64"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
run tower_1 $B
run tower_2 $B
run empty $B --scenario Empty
run tower4096 $B --envs-per-gpu 4096
run driver_1 $B --gpus 1 --steps 20 --warmup 5
run driver_2 $B --gpus 1 --steps 20 --warmup 5
run tower128x72 $B --obs 128 72
