cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c2; mkdir -p $O
cd $R; python bench.py --agents 4 --envs-per-gpu 512 --no-cpu-baseline > $O/tower_512x4_bench.json 2> $O/err.txt
cd /tmp; rocprofv3 --kernel-trace --stats -d $O/db1 -o run -- python $R/bench.py --agents 4 --envs-per-gpu 512 --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $O/l1.log 2>&1
python $R/scripts/rocpd_summary.py $O/db1/run_results.db > $O/tower_512x4_kernel_stats.csv
MV_PIPELINE=0 rocprofv3 --kernel-trace --stats -d $O/db2 -o run -- python $R/bench.py --agents 4 --envs-per-gpu 512 --steps 400 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $O/l2.log 2>&1
python $R/scripts/rocpd_summary.py $O/db2/run_results.db > $O/tower_512x4_unpipelined_kernel_stats.csv
rm -rf $O/db1 $O/db2
cd $R; python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-extra-legs --profile-steps 32 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('headline', round(l['value']/1e6,2), round(l['roofline_physics']['avg_launch_ms']*1e3,1))"
sed -n 3,4p $O/tower_512x4_unpipelined_kernel_stats.csv | cut -c1-120
tail -1 $O/tower_512x4_bench.json | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(round(l['value']/1e6,2), {k[6:]:round(v/1e6,2) for k,v in l.items() if k.startswith('value_')})"
