#!/bin/bash
# after scripts/make_pmc_traffic.py r09z 16: the default bench line and the driver's form with roofline.traffic / roofline.valu / roofline.lds read back from
# profiles/pmc_traffic.json (same kernel sources: the hash guard), BASELINE.md section 3's CPU baseline in full, and the tests added after the final suite ran
set -u
TAG=${1:-r09zc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_env_surface_gpu.py tests/test_abi.py -q > $OUT/pytest_env_surface.log 2>&1; tail -3 $OUT/pytest_env_surface.log
timeout 900 python bench.py > $OUT/tower_bench_with_counters.json 2> $OUT/tower_bench_with_counters.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style_with_counters.json 2> $OUT/tower_bench_driver_style_with_counters.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style_with_counters_2.json 2> /dev/null
for f in $OUT/tower_bench_with_counters.json $OUT/tower_bench_driver_style_with_counters.json $OUT/tower_bench_driver_style_with_counters_2.json; do python -c "import json; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['traffic'], round(d['roofline']['frac'],3), d['roofline'].get('lds',{}).get('conflict_frac'), d['roofline'].get('valu',{}).get('insts_per_launch'))"; done
timeout 1500 python bench.py --cpu-baseline-full --no-extra-legs > $OUT/tower_bench_cpu_baseline_full.json 2> $OUT/tower_bench_cpu_baseline_full.err
python -c "import json; d=json.load(open('$OUT/tower_bench_cpu_baseline_full.json')); print('cpu baseline full', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:200])"
