#!/bin/bash
# round 3, last pass (after the r03c profile: shared-out controllers, simDone on the dispatch packet, box corners as sums, records loaded beside the header):
# headline bench (all legs + CPU baseline), the driver's short form, kernel tables pipelined and unpipelined
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R; timeout 600 python bench.py > $O/tower_bench.json 2> $O/tower_bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/tower_bench_driver_style.json 2> $O/tower_bench_driver_style.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/db1 -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $O/l1.log 2>&1
python $R/scripts/rocpd_summary.py $O/db1/run_results.db > $O/tower_kernel_stats.csv
MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/db2 -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $O/l2.log 2>&1
python $R/scripts/rocpd_summary.py $O/db2/run_results.db > $O/tower_unpipelined_kernel_stats.csv
rm -rf $O/db1 $O/db2
for f in tower_bench tower_bench_driver_style; do tail -1 $O/$f.json | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$f', round(l['value']/1e6,2), round(l['ms_per_step']*1e3,2), {k[6:]:round(v/1e6,2) for k,v in l.items() if k.startswith('value_')})"; done
sed -n 3,4p $O/tower_kernel_stats.csv | cut -c1-120; sed -n 3,4p $O/tower_unpipelined_kernel_stats.csv | cut -c1-120
