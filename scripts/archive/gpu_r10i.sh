#!/bin/bash
# r10i: the long-list pass with 32 x 2-pixel tiles per wave (x 2 rows of them with two pixels per lane: every row a wave stores is one whole 128-byte line;
# -DMV_GLIST_TILE_W=32) against 16 x 4: parity tests on the variant, Collect / HexMemory / HexExplore / Mixed, WRITE_SIZE of the pass
set -u
TAG=${1:-r10i}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_gt32.so timeout 1800 python -m pytest tests/test_hex_parity_gpu.py tests/test_collect_parity_gpu.py tests/test_fast_pixels_gpu.py tests/test_multitask_gpu.py -m gpu -q -x > $OUT/pytest_gt32.log 2>&1; echo "rc=$?" >> $OUT/pytest_gt32.log
tail -3 $OUT/pytest_gt32.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  for V in t16 gt32; do
    LIB=""; [ $V = gt32 ] && LIB=$R/megaverse_amd/_variants/libmv_gt32.so
    MV_LIB_PATH=$LIB run hexmemory_${V}_$i $B --scenario HexMemory
    MV_LIB_PATH=$LIB run collect_${V}_$i $B --scenario Collect
    MV_LIB_PATH=$LIB run hexexplore_${V}_$i $B --scenario HexExplore
    MV_LIB_PATH=$LIB run mixed64_${V}_$i $B --scenario Mixed --obs 64 64
    MV_LIB_PATH=$LIB run collect72_${V}_$i $B --scenario Collect --obs 128 72
  done
done
for V in t16 gt32; do
  LIB=""; [ $V = gt32 ] && LIB=$R/megaverse_amd/_variants/libmv_gt32.so
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" "WRITE_SIZE"; do
    T=$(echo $C | cut -d' ' -f1)
    (cd /tmp; MV_LIB_PATH=$LIB timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/db_$V -o run -- python $R/bench.py --scenario Collect --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc_Collect_${V}_$T.log 2>&1
     python $R/scripts/rocpd_summary.py $OUT/db_$V/run_results.db --pmc > $OUT/pmc_Collect_${V}_$T.csv 2>> $OUT/pmc_Collect_${V}_$T.log; rm -rf $OUT/db_$V)
    echo "== Collect $V $T"; grep -h "raster_glist_batch" $OUT/pmc_Collect_${V}_$T.csv | cut -c1-40,100-200
  done
done
