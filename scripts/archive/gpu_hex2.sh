#!/bin/bash
# gpu_hex2.sh <outdir>: fast-pixel tests of the hex scenarios + hex benches
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_fast_pixels_gpu.py -k "Hex" -m gpu -q -x -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 400 --warmup 50 --no-cpu-baseline"
$B --scenario HexMemory > $OUT/bench_hexmemory.json 2>&1
$B --scenario HexExplore > $OUT/bench_hexexplore.json 2>&1
$B --scenario HexMemory --agents 4 --envs-per-gpu 256 > $OUT/bench_hexmemory_a4.json 2>&1
$B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2>&1
tail -14 $OUT/pytest.log
for f in $OUT/bench_*.json; do echo $f; tail -1 $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,3),'M obs/s', round(d['ms_per_step'],4),'ms', 'raster', d.get('roofline',{}).get('avg_launch_ms'), 'step', d.get('roofline_physics',{}).get('avg_launch_ms'))"; done
