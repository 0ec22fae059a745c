#!/bin/bash
# r05g: wave priority by remaining work (longest remaining work first)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05g; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
for cfg in "lrtf0 MV_RASTER_LRTF=0" "lrtf1 MV_RASTER_LRTF=1" "lrtf1_notail MV_RASTER_LRTF=1 MV_RASTER_TAIL_DIV=0"; do
  set -- $cfg; N=$1; shift
  env "$@" MV_PIPELINE=0 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_rtiming.so timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/rt_$N.json 2> $OUT/rt_$N.err
  echo "== $N"; grep "raster timing (\|life by decile" $OUT/rt_$N.err | cut -c1-330
  (cd /tmp; env "$@" MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/u.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/ks_$N.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
  grep -h "raster_fast" $OUT/ks_$N.csv | cut -d, -f3-8
  env "$@" timeout 300 python bench.py --no-cpu-baseline > $OUT/tower_$N.json 2> $OUT/tower_$N.err
  python -c "import json; d=json.load(open('$OUT/tower_$N.json')); print(round(d['value']/1e6,2), 'M', {k: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['avg_launch_ms'])"
done
