#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_multitask_gpu.py tests/test_full_size_gpu.py -k "multitask or mixed" -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
B="python bench.py --steps 1000 --warmup 50 --no-cpu-baseline"
$B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64.json 2>&1
$B --scenario Mixed > $OUT/bench_mixed128.json 2>&1
tail -4 $OUT/pytest.log
for f in $OUT/bench_*.json; do echo $f; tail -1 $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']/1e6,3),'M obs/s', round(d['ms_per_step'],4),'ms')"; done
