#!/bin/bash
# r10d: the long-list pass finds a tile's first round with primitives before it sets the rays up (empty tiles cleared at once, no zero-initialised registers) against round 5's body (base)
set -u
TAG=${1:-r10d}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1800 python -m pytest tests/test_hex_parity_gpu.py tests/test_collect_parity_gpu.py tests/test_fast_pixels_gpu.py tests/test_multitask_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for V in new base; do
    LIB=""; [ $V = base ] && LIB=$R/megaverse_amd/_variants/libmv_base.so
    MV_LIB_PATH=$LIB run hexmemory_${V}_$i $B --scenario HexMemory
    MV_LIB_PATH=$LIB run collect_${V}_$i $B --scenario Collect
    MV_LIB_PATH=$LIB run hexexplore_${V}_$i $B --scenario HexExplore
    MV_LIB_PATH=$LIB run mixed64_${V}_$i $B --scenario Mixed --obs 64 64
  done
done
for V in new base; do
  LIB=""; [ $V = base ] && LIB=$R/megaverse_amd/_variants/libmv_base.so
  for S in HexMemory Collect; do
    (cd /tmp; MV_LIB_PATH=$LIB timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/db_${S}_$V -o run -- python $R/bench.py --scenario $S --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc_${S}_$V.log 2>&1
     python $R/scripts/rocpd_summary.py $OUT/db_${S}_$V/run_results.db --pmc > $OUT/pmc_${S}_$V.csv 2>> $OUT/pmc_${S}_$V.log; rm -rf $OUT/db_${S}_$V)
    echo "== $S $V"; grep -h "raster_glist_batch" $OUT/pmc_${S}_$V.csv | cut -c1-40,100-200
  done
done
