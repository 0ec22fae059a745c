#!/bin/bash
# round 3, fourth GPU pass: where the Mixed observation time goes (kernel trace), raster split sweep at 64x64
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03d}
mkdir -p $OUT
cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
B="python bench.py --steps 600 --warmup 50 --no-cpu-baseline --profile-steps 64 --no-extra-legs"
for sp in 1 2 4; do
  MV_RASTER_SPLIT=$sp timeout 300 $B --scenario Mixed --obs 64 64 > $OUT/bench_mixed64_split$sp.json 2> $OUT/bench_mixed64_split$sp.err
done
for sc in HexMemory Collect TowerBuilding; do
  for sp in 1 2 4; do
    MV_RASTER_SPLIT=$sp timeout 300 $B --scenario $sc --obs 64 64 > $OUT/bench_${sc}64_split$sp.json 2> $OUT/bench_${sc}64_split$sp.err
  done
done
MV_RASTER_SPLIT=2 timeout 300 $B --scenario Mixed --obs 128 128 > $OUT/bench_mixed128_split2.json 2> $OUT/bench_mixed128_split2.err
cd /tmp
for sp in 1 4; do
MV_RASTER_SPLIT=$sp MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_mixed_$sp -o run -- python $R/bench.py --scenario Mixed --obs 64 64 --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/mixed_stats_$sp.log 2>&1
python $R/scripts/rocpd_summary.py $OUT/db_mixed_$sp/run_results.db > $OUT/mixed64_unpipelined_split${sp}_kernel_stats.csv 2>> $OUT/mixed_stats_$sp.log
done
rm -rf $OUT/db_*
cd $R
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        l=json.loads(open(f).read().strip().splitlines()[-1])
        r=l.get("roofline",{}); p=l.get("roofline_physics",{})
        print(os.path.basename(f), "%.2fM %.4fms"%(l["value"]/1e6,l["ms_per_step"]), "raster %.4f step %.4f"%(r.get("avg_launch_ms",0),p.get("avg_launch_ms",0)))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json",".err")).read()[-600:])
PY
head -8 $OUT/mixed64_unpipelined_split*_kernel_stats.csv
