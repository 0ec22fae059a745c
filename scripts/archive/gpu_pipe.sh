#!/bin/bash
# gpu_pipe.sh <outdir>: pipelining tests + a slice of the parity suite + benches
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_pipelining_gpu.py tests/test_fast_pixels_gpu.py tests/test_refill_protocol_gpu.py tests/test_env_surface_gpu.py tests/test_multitask_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline"
$B > $OUT/bench_tower.json 2>&1
$B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4.json 2>&1
$B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_obsthard512.json 2>&1
$B --scenario Collect > $OUT/bench_collect.json 2>&1
$B --scenario HexMemory > $OUT/bench_hexmemory.json 2>&1
$B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2>&1
$B --pixels exact > $OUT/bench_tower_exact.json 2>&1
MV_FAST_WAVES=6 $B > $OUT/bench_tower_w6.json 2>&1
MV_PIPELINE=0 $B > $OUT/bench_tower_nopipe.json 2>&1
MV_PIPELINE=0 $B --scenario ObstaclesHard --envs-per-gpu 512 > $OUT/bench_obsthard512_nopipe.json 2>&1
MV_FAST_WAVES=6 $B --agents 4 --envs-per-gpu 512 > $OUT/bench_a4_w6.json 2>&1
tail -15 $OUT/pytest.log
for f in $OUT/bench_*.json; do echo $f; tail -1 $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,3),'M obs/s', round(d['ms_per_step'],4),'ms', 'raster', d.get('roofline',{}).get('avg_launch_ms'), 'step', d.get('roofline_physics',{}).get('avg_launch_ms'))"; done
