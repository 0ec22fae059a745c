#!/bin/bash
# r10n: MegaverseEnv.step_device (three device tensors, no host synchronisation) beside step_batched in the bench's env legs; its test
set -u
TAG=${1:-r10n}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_env_surface_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline --profile-steps 64 --steps 400 > $OUT/tower_env_legs.json 2> $OUT/tower_env_legs.err
python -c "
import json; d=json.loads(open('$OUT/tower_env_legs.json').read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})"
