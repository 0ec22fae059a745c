#!/bin/bash
# (mv_set_pass_stream and the probe's PASS_STREAM switch were built for this measurement and not kept)
# r10e: double-buffered sampling with the halves' observation passes on ONE stream of the caller's (mv_set_pass_stream; PASS_STREAM=1) against each half
# on its own stream (the chains fall into step: r03c timeline), scripts/probe_double_buffer.py, 2 x 512 envs; timeline of the new form
set -u
TAG=${1:-r10e}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_pipelining_gpu.py -m gpu -q -x -k "taking_turns or policy_in_the_loop" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for i in 1 2; do
  for P in 0 1; do
    PASS_STREAM=$P timeout 300 python scripts/probe_double_buffer.py 1024 2000 > $OUT/double_buffer_pass_stream${P}_$i.txt 2>&1; tail -1 $OUT/double_buffer_pass_stream${P}_$i.txt
  done
done
PASS_STREAM=1 timeout 300 python scripts/probe_double_buffer.py 2048 1000 > $OUT/double_buffer_pass_stream1_2048.txt 2>&1; tail -1 $OUT/double_buffer_pass_stream1_2048.txt
PASS_STREAM=0 timeout 300 python scripts/probe_double_buffer.py 2048 1000 > $OUT/double_buffer_pass_stream0_2048.txt 2>&1; tail -1 $OUT/double_buffer_pass_stream0_2048.txt
(cd /tmp; PASS_STREAM=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t -o run -- python $R/scripts/probe_double_buffer.py 1024 200 > $OUT/trace.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_t/run_results.db 60 40 > $OUT/timeline_double_buffered_pass_stream.txt 2>/dev/null; rm -rf $OUT/db_t)
head -40 $OUT/timeline_double_buffered_pass_stream.txt | cut -c1-100
