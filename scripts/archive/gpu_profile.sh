#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + PMC passes.
# Output: gpurun_out/{bench.json, prof_stats/, pmc_*/}; summaries get copied into profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 400 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o run -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
echo "stats rc=$?" >> $OUT/prof_stats.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc_$C.log 2>&1
  echo "pmc $C rc=$?" >> $OUT/pmc_$C.log
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --kernel-trace -d $OUT/pmc_SQ -o run -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 > $OUT/pmc_SQ.log 2>&1
echo "pmc SQ rc=$?" >> $OUT/pmc_SQ.log
find $OUT -name "*.csv" | head -50 > $OUT/csv_list.txt
# keep the merge small: drop big traces, keep stats + counter csvs
find $OUT -name "*.db" -size +20M -delete
du -sh $OUT >> $OUT/csv_list.txt
# other scenario families: kernel stats only
cd /tmp
for S in "ObstaclesHard --envs-per-gpu 512" "Collect" "Rearrange"; do
  N=$(echo $S | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats_$N -o run -- python $R/bench.py --scenario $S --steps 400 --warmup 50 --no-cpu-baseline > $OUT/prof_stats_$N.log 2>&1
  cd $R; timeout 200 python bench.py --scenario $S > $OUT/bench_$N.json 2> $OUT/bench_$N.err; cd /tmp
done
find $OUT -name "*.db" -size +20M -delete
