#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04z; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 1800 python -m pytest tests/test_full_size_oracle_gpu.py tests/test_pipelining_gpu.py tests/test_canonical_poses_gpu.py -x -q 2>&1 | tail -6
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 0"
for a in "" "--agents 4 --envs-per-gpu 512" "--agents 2 --envs-per-gpu 512"; do echo "$a: $($B $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))")"; done
