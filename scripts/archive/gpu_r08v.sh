#!/bin/bash
# r08v: the double-buffered policy-in-the-loop leg with one host thread per half (bench.py), draw waits once per draw launch; pipelining / parity tests
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r08v; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_pipelining_gpu.py tests/test_parity_gpu.py tests/test_env_surface_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 > $OUT/tower_${i}_bench.json 2> $OUT/tower_${i}.err
done
timeout 300 python bench.py --no-cpu-baseline --profile-steps 0 --envs-per-gpu 512 --agents 4 > $OUT/tower_512x4_bench.json 2> /dev/null
for f in $OUT/*_bench.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,2), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, {k[22:]: round(v,4) for k,v in d.items() if k.startswith('host_enqueue')})
"; done
tail -3 $OUT/tower_1.err
