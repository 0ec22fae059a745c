#!/bin/bash
# the last change (batched group calls up to 1024 envs): multitask / pipelining tests, Mixed at 1024 and 2048 envs, then the PMC passes + bench lines of the committed sources
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08z_last; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban TMPDIR=/tmp
timeout 900 python -m pytest tests/test_multitask_gpu.py tests/test_pipelining_gpu.py tests/test_refill_protocol_gpu.py -q -m gpu 2>&1 | tail -2
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
$B --scenario Mixed --obs 64 64 > $OUT/mixed_64_bench.json 2> /dev/null
$B --scenario Mixed --obs 64 64 --envs-per-gpu 2048 > $OUT/mixed_64_2048_bench.json 2> /dev/null
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), d['config'].get('launches_per_call'))
"; done
bash scripts/gpu_r08_pmc.sh r08z
