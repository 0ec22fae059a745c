#!/bin/bash
# few-frame latency of the long-list raster: split sweep on 128 HexMemory / Collect envs at 64 x 64 and 128 x 128 (kernel trace, unpipelined)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r03i}
mkdir -p $OUT
cd /tmp
for sc in HexMemory Collect; do
 for ob in 64 128; do
  for sp in 4 8 16; do
   MV_RASTER_SPLIT=$sp MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db -o run -- python $R/bench.py --scenario $sc --envs-per-gpu 128 --obs $ob $ob --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/log.txt 2>&1
   echo "$sc $ob split $sp: $(python $R/scripts/rocpd_summary.py $OUT/db/run_results.db | grep -E 'raster|step' | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print(r[0][9:34], 'avg_us', round(float(r[3])/1000,1), end=' | ')")"
   rm -rf $OUT/db
  done
 done
done | tee $OUT/split_sweep.txt
