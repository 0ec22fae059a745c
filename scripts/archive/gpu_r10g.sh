#!/bin/bash
# r10g: Collect's generator after the noise rewrite (per-axis tables, branch-free gradient corners; the same bytes): episodes per second and thread on the GPU
# box, and Collect / Mixed with all host cores and with two (taskset -c 0,1 = a 16-CPU quota / 8 ranks; r09e before: Collect 14.2 / 9.0 M obs/s)
set -u
TAG=${1:-r10g}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
taskset -c 0 python scripts/probe_generators.py 400 > $OUT/probe_generators_gpu_box.txt 2>&1; cat $OUT/probe_generators_gpu_box.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3))
except Exception as e: print('$name', 'failed', e)
PY
}
for i in 1 2; do
  run Collect_all_cores_$i $B --scenario Collect
  run Collect_2_cores_$i taskset -c 0,1 $B --scenario Collect
done
run Collect72_all_cores $B --scenario Collect --obs 128 72
run Collect72_2_cores taskset -c 0,1 $B --scenario Collect --obs 128 72
run mixed64_all_cores $B --scenario Mixed --obs 64 64
run mixed64_2_cores taskset -c 0,1 $B --scenario Mixed --obs 64 64
