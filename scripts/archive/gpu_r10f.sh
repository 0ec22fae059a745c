#!/bin/bash
# (mv_set_pass_stream and the probe's PASS_STREAM switch were built for this measurement and not kept)
# r10f: r10e's double-buffered probe with more hardware queues (GPU_MAX_HW_QUEUES: HIP deals its streams over 4 by default -- r10e's timeline shows the two halves'
# policy + step streams on ONE queue, each half's wait for its pass blocking the other half behind it)
set -u
TAG=${1:-r10f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for Q in 4 8 16; do
  for P in 0 1; do
    GPU_MAX_HW_QUEUES=$Q PASS_STREAM=$P timeout 300 python scripts/probe_double_buffer.py 1024 2000 > $OUT/double_buffer_q${Q}_pass_stream${P}.txt 2>&1; echo "queues $Q pass stream $P: $(tail -1 $OUT/double_buffer_q${Q}_pass_stream${P}.txt)"
  done
done
(cd /tmp; GPU_MAX_HW_QUEUES=16 PASS_STREAM=1 timeout 300 rocprofv3 --kernel-trace -d $OUT/db_t -o run -- python $R/scripts/probe_double_buffer.py 1024 200 > $OUT/trace.log 2>&1
 python $R/scripts/kernel_timeline.py $OUT/db_t/run_results.db 60 40 > $OUT/timeline_double_buffered_pass_stream_16_queues.txt 2>/dev/null; rm -rf $OUT/db_t)
head -40 $OUT/timeline_double_buffered_pass_stream_16_queues.txt | cut -c1-100
