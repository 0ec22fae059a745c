#!/bin/bash
# r09h: where does a Hex / Collect step launch's time go (-DMV_TICK_TIMING builds: frame setup phases, printed by mv_close)?  + the hex tick's cluster pre-test
set -u
TAG=${1:-r09h}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests/test_hex_parity_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
for s in HexMemory Collect; do
  MV_TICK_TIMING=1 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_ticktime.so $B --scenario $s --steps 300 --warmup 50 > $OUT/${s}_timing.json 2> $OUT/${s}_timing.err
  grep "mv tick timing" $OUT/${s}_timing.err | cut -c1-200
done
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
run HexMemory $B --scenario HexMemory
MV_BOX_CLUSTERS=0 run HexMemory_c0 $B --scenario HexMemory
MV_PIPELINE=0 run HexMemory_alone $B --scenario HexMemory --steps 400
MV_PIPELINE=0 MV_BOX_CLUSTERS=0 run HexMemory_alone_c0 $B --scenario HexMemory --steps 400
