#!/bin/bash
# r09a: the software-pipelined multi-tick step kernels (two waves per env: tick j + 1 beside frame setup j) against the one-wave kernels (MV_STEP_PIPE=0)
set -u
TAG=${1:-r09a}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/test_pipelining_gpu.py tests/test_full_size_oracle_gpu.py tests/test_refill_protocol_gpu.py tests/test_obstacles_parity_gpu.py tests/test_empty_parity_gpu.py tests/test_parity_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for P in 1 0; do
  export MV_STEP_PIPE=$P
  run tower_p$P $B
  run tower_p${P}_b $B
  MV_PIPELINE=0 run tower_alone_p$P $B --steps 400
  run tower_512_p$P $B --envs-per-gpu 512
  run tower_4096_p$P $B --envs-per-gpu 4096
  run empty_p$P $B --scenario Empty
  MV_PIPELINE=0 run empty_alone_p$P $B --scenario Empty --steps 400
  run obst_hard_512_p$P $B --scenario ObstaclesHard --envs-per-gpu 512
  run obst_hard_512_noov_p$P $B --scenario ObstaclesHard --envs-per-gpu 512 --pass-overlap off
  run obst_hard_1024_p$P $B --scenario ObstaclesHard
done
