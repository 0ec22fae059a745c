#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06z; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
run() { # name, env..., -- args
  N=$1; shift; E=(); while [ "$1" != "--" ]; do E+=("$1"); shift; done; shift
  env "${E[@]}" timeout 300 python bench.py "$@" --steps 3200 --warmup 400 --no-cpu-baseline --profile-steps 100 --no-extra-legs > $OUT/$N.json 2> $OUT/$N.err
  python - $OUT/$N.json $N <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.2f M"%(d["value"]/1e6), "raster/tick %.1f us"%(d["roofline"]["avg_launch_ms"]*1e3), "step/tick %.1f us"%(d["roofline_physics"]["avg_launch_ms"]*1e3))
PY
}
run tw1024 X=1 -- --envs-per-gpu 1024
run tw1024u X=1 MV_PIPELINE=0 -- --envs-per-gpu 1024
run tw512 X=1 -- --envs-per-gpu 512
run x4_w3 X=1 -- --envs-per-gpu 512 --agents 4
run x4_w4 MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_ag4.so -- --envs-per-gpu 512 --agents 4
run b1 X=1 -- --batch 1
run tw4096 X=1 -- --envs-per-gpu 4096
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_reset_parity_gpu.py tests/test_pipelining_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log | cut -c1-300
