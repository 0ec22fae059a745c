#!/bin/bash
# r09c: the classified frames' tiles in class order from a compact list + the empty tiles cleared by the whole workgroup (MV_TILE_LIST=0 variant: frame order, empty tiles handed out)
set -u
TAG=${1:-r09c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests/test_fast_pixels_gpu.py tests/test_canonical_frames_gpu.py tests/test_parity_gpu.py tests/test_py_surface_gpu.py tests/test_full_size_gpu.py tests/test_multitask_gpu.py tests/test_rearrange_parity_gpu.py tests/test_sokoban_parity_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
export MV_STEP_PIPE=0
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for V in list nolist; do
  if [ $V = nolist ]; then export MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_nolist.so; else unset MV_LIB_PATH; fi
  run tower_${V}_1 $B
  run tower_${V}_2 $B
  run empty_${V} $B --scenario Empty
  run tower512_${V} $B --envs-per-gpu 512
  run tower4096_${V} $B --envs-per-gpu 4096
  run tower512x4_${V} $B --envs-per-gpu 512 --agents 4
  run obsthard_${V} $B --scenario ObstaclesHard
  run rearrange_${V} $B --scenario Rearrange
  run sokoban_${V} $B --scenario Sokoban
  run driver_${V} timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 16
  (cd /tmp; MV_BENCH_CALL_SCHEDULE=16 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/db_$V -o run -- python $R/bench.py --batch 16 --steps 128 --warmup 32 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc_$V.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_$V/run_results.db --pmc > $OUT/pmc_$V.csv 2>> $OUT/pmc_$V.log; rm -rf $OUT/db_$V)
  grep -h "raster_fast_batch" $OUT/pmc_$V.csv | cut -c1-200
done
