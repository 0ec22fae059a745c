#!/bin/bash
# r08y2: confidence in the settled refill protocol: a long soak, the run-ahead test five times over, the refill / pipelining tests twice
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08y2; mkdir -p $OUT; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 1500 python scripts/soak.py 24000 > $OUT/soak_24000.log 2>&1; tail -3 $OUT/soak_24000.log
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_refill_protocol_gpu.py -q -m gpu -k far_ahead 2>&1 | tail -1; done
for i in 1 2; do timeout 600 python -m pytest tests/test_refill_protocol_gpu.py tests/test_pipelining_gpu.py tests/test_multitask_gpu.py -q -m gpu 2>&1 | tail -1; done
