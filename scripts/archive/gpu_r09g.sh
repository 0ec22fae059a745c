#!/bin/bash
# r09g: box clusters (long lists: the frame setup skips the clusters of 64 static boxes the camera cannot see); MV_BOX_CLUSTERS=0: as before
set -u
TAG=${1:-r09g}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 2400 python -m pytest tests/test_hex_parity_gpu.py tests/test_collect_parity_gpu.py tests/test_multitask_gpu.py tests/test_fast_pixels_gpu.py tests/test_soak_gpu.py tests/test_full_size_oracle_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for C in 1 0; do
  export MV_BOX_CLUSTERS=$C
  for s in HexMemory HexExplore Collect; do run ${s}_c$C $B --scenario $s; done
  MV_PIPELINE=0 run HexMemory_alone_c$C $B --scenario HexMemory --steps 400
  MV_PIPELINE=0 run Collect_alone_c$C $B --scenario Collect --steps 400
  run mixed64_c${C}_1 $B --scenario Mixed --obs 64 64
  run mixed64_c${C}_2 $B --scenario Mixed --obs 64 64
  run mixed4_64_c${C} $B --scenario Mixed4 --obs 64 64
done
unset MV_BOX_CLUSTERS
run mixed128 $B --scenario Mixed --obs 128 128
run tower64 $B --obs 64 64
run tower128x72 $B --obs 128 72
run collect128x72 $B --scenario Collect --obs 128 72
