#!/bin/bash
# r04t: does the second round of workgroups cost what the timeline suggests?  raster alone at 768 / 896 / 1024 / 1152 frames (1792 workgroup slots)
set -u
TAG=${1:-r04t}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for N in 768 896 1024 1152 1280; do
  (cd /tmp; MV_PIPELINE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/db_u -o run -- python $R/bench.py --envs-per-gpu $N --steps 300 --warmup 50 --no-cpu-baseline --profile-steps 0 --no-extra-legs --batch 1 > $OUT/u.log 2>&1
   python $R/scripts/rocpd_summary.py $OUT/db_u/run_results.db > $OUT/tower_unpipelined_n${N}_kernel_stats.csv 2>> $OUT/u.log; rm -rf $OUT/db_u)
  echo "N=$N: $(grep -h raster_fast $OUT/tower_unpipelined_n${N}_kernel_stats.csv | awk -F, -v n=$N '{printf "%.1f us per launch, %.2f ns per frame", $(NF-3)/1000, $(NF-3)/n}')"
done
