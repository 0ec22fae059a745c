#!/bin/bash
# r09f: configs[4] (Mixed 64 x 64): the short-list gyms' frames of the union observation launch with the tile classification (and whole frames per workgroup: the
# classification needs >= 32 tiles), the long-list gyms' step workgroups with two waves instead of four
set -u
TAG=${1:-r09f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for S in Mixed Mixed4; do
  run ${S}_base_1 $B --scenario $S --obs 64 64
  run ${S}_base_2 $B --scenario $S --obs 64 64
  for SP in 1 2; do MV_UNION_SPLIT_SMALL=$SP run ${S}_split$SP $B --scenario $S --obs 64 64; done
  for SP in 1 2 4; do MV_UNION_SPLIT_SMALL=$SP MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_unioncls.so run ${S}_cls_split$SP $B --scenario $S --obs 64 64; done
  MV_UNION_TICKS_WAVES=2 run ${S}_stepwaves2 $B --scenario $S --obs 64 64
  MV_UNION_TICKS_WAVES=2 MV_UNION_SPLIT_SMALL=1 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_unioncls.so run ${S}_cls_split1_stepwaves2 $B --scenario $S --obs 64 64
done
run Mixed128_base $B --scenario Mixed --obs 128 128
MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_unioncls.so run Mixed128_cls $B --scenario Mixed --obs 128 128
MV_UNION_SPLIT_SMALL=1 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_unioncls.so run Mixed128_cls_split1 $B --scenario Mixed --obs 128 128
MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_unioncls.so timeout 900 python -m pytest tests/test_multitask_gpu.py tests/test_full_size_oracle_gpu.py -m gpu -q -k "mixed or multitask or union or grouped" > $OUT/pytest_cls.log 2>&1; tail -3 $OUT/pytest_cls.log
