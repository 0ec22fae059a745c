#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for cfg in "oh512 --scenario ObstaclesHard --envs-per-gpu 512" "tw1024 --envs-per-gpu 1024" "tw512 --envs-per-gpu 512"; do
  set -- $cfg; N=$1; shift
  timeout 300 rocprofv3 --kernel-trace -d $OUT/db_$N -o run -- python $R/bench.py "$@" --steps 400 --warmup 80 --no-cpu-baseline --no-extra-legs --profile-steps 0 > $OUT/timeline_$N.log 2>&1
  python $R/scripts/kernel_timeline.py $OUT/db_$N/run_results.db 24 10 > $OUT/timeline_$N.txt 2>> $OUT/timeline_$N.log; rm -rf $OUT/db_$N
  echo "== $N"; cat $OUT/timeline_$N.txt | cut -c1-110
done
