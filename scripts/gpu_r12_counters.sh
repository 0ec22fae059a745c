#!/bin/bash
# after scripts/make_pmc_traffic.py r12z 16: the default bench line and the driver's form with roofline.traffic / roofline.valu / roofline.lds read back from profiles/pmc_traffic.json (same kernel sources: the hash guard)
set -u
TAG=${1:-r12zc}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/tower_bench_with_counters.json 2> $OUT/tower_bench_with_counters.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/tower_bench_driver_style_with_counters.json 2> $OUT/tower_bench_driver_style_with_counters.err
for f in $OUT/tower_bench_with_counters.json $OUT/tower_bench_driver_style_with_counters.json; do python -c "import json; d=json.load(open('$f')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['traffic'], round(d['roofline']['frac'],3), d['roofline'].get('lds',{}).get('conflict_frac'), d['roofline'].get('valu',{}).get('insts_per_launch'), d['config'].get('host_generator_threads'))"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
