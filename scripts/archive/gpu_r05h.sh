#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05h; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
export BOXOBAN_LEVELS=$R/tests/golden/boxoban
timeout 900 python -m pytest tests/test_fast_pixels_gpu.py tests/test_multitask_gpu.py -x -q -k "Collect or Hex or multitask or union" 2>&1 | tail -3
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs"
for sc in Collect HexMemory HexExplore; do echo "$sc: $($B --scenario $sc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), d['roofline']['avg_launch_ms'], d['roofline_physics']['avg_launch_ms'])")"; done
echo "mixed64: $($B --scenario Mixed --obs 64 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4))")"
