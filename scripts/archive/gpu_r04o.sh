#!/bin/bash
# r04o: the whole GPU suite on the new code, then the headline legs
set -u
TAG=${1:-r04o}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/tower_bench.json 2> $OUT/tower_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style.json 2> $OUT/tower_bench_driver_style.err
for f in $OUT/*.json; do echo "$(basename $f): $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), {k: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['avg_launch_ms'], d['roofline_physics']['avg_launch_ms'])" 2>/dev/null)"; done
