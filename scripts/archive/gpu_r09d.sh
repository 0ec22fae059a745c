#!/bin/bash
# r09d: planar / empty RUNS in the drawing order (up to four neighbours of a tile row per entry); the long-list step kernels at 96 VGPRs; the pipelined Obstacles step kernel at 168
set -u
TAG=${1:-r09d}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests/test_fast_pixels_gpu.py tests/test_canonical_frames_gpu.py tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_multitask_gpu.py tests/test_rearrange_parity_gpu.py tests/test_sokoban_parity_gpu.py tests/test_empty_parity_gpu.py tests/test_obstacles_parity_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
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
run tower_1 $B
run tower_2 $B
run empty $B --scenario Empty
MV_STEP_PIPE=1 run empty_pipe $B --scenario Empty
MV_STEP_PIPE=1 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_obst3.so run empty_pipe_168 $B --scenario Empty
run tower512 $B --envs-per-gpu 512
run tower4096 $B --envs-per-gpu 4096
run tower512x4 $B --envs-per-gpu 512 --agents 4
run obsthard $B --scenario ObstaclesHard
MV_STEP_PIPE=1 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_obst3.so run obsthard_pipe_168 $B --scenario ObstaclesHard
run obsthard512 $B --scenario ObstaclesHard --envs-per-gpu 512
MV_STEP_PIPE=1 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_obst3.so run obsthard512_pipe_168 $B --scenario ObstaclesHard --envs-per-gpu 512
run rearrange $B --scenario Rearrange
run sokoban $B --scenario Sokoban
run driver timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 16
for s in HexMemory Collect HexExplore; do
  run ${s} $B --scenario $s
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_hex5.so run ${s}_96vgpr $B --scenario $s
done
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
run mixed64 $B --scenario Mixed --obs 64 64
run mixed4_64 $B --scenario Mixed4 --obs 64 64
(cd /tmp; MV_BENCH_CALL_SCHEDULE=16 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/db -o run -- python $R/bench.py --batch 16 --steps 128 --warmup 32 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pmc.log 2>&1
 python $R/scripts/rocpd_summary.py $OUT/db/run_results.db --pmc > $OUT/pmc.csv 2>> $OUT/pmc.log; rm -rf $OUT/db)
grep -h "raster_fast_batch" $OUT/pmc.csv | cut -c1-200
