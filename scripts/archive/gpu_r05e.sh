#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05e; mkdir -p $OUT; export TMPDIR=/tmp; cd $R
timeout 1200 python -m pytest tests/test_fast_pixels_gpu.py tests/test_pipelining_gpu.py tests/test_canonical_frames_gpu.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline > $OUT/tower.json 2> $OUT/tower.err
python -c "import json; d=json.load(open('$OUT/tower.json')); print(round(d['value']/1e6,2), 'M', {k: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')}, d['roofline']['avg_launch_ms'], d['roofline_physics']['avg_launch_ms'])"
MV_RASTER_TAIL_DIV=0 timeout 600 python bench.py --no-cpu-baseline > $OUT/tower_notail.json 2> $OUT/tower_notail.err
python -c "import json; d=json.load(open('$OUT/tower_notail.json')); print('no tail split:', round(d['value']/1e6,2), 'M', {k: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})"
