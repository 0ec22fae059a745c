#!/bin/bash
# r09b: the new parity tests (bench launch shapes at full size, mixed batched, soak, goal-terminated refills, batched gather), the env_step legs, the hist_done fence A/B,
# the step launch beside a pass with a smaller code footprint (MV_PLANAR=0 / 3: is the slowdown beside the passes instruction-cache pressure?)
set -u
TAG=${1:-r09b}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 2700 python -m pytest tests/test_full_size_oracle_gpu.py tests/test_soak_gpu.py tests/test_refill_protocol_gpu.py tests/test_distributed_gpu.py tests/test_multitask_gpu.py tests/test_py_surface_gpu.py tests/test_env_surface_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --profile-steps 128"
run() { local name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - <<PY
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value']/1e6,2), 'M', 'raster/tick %.1f us step/tick %.1f us' % (d['roofline']['avg_launch_ms']*1e3, d['roofline_physics']['avg_launch_ms']*1e3), {k[6:]: round(v/1e6,2) for k,v in d.items() if k.startswith('value_')})
except Exception as e: print('$name', 'failed', e)
PY
}
export MV_STEP_PIPE=0
run tower_full timeout 600 python bench.py --no-cpu-baseline
for P in 0 1; do
  MV_STEP_PIPE=$P run driver_p${P}_a timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 0
  MV_STEP_PIPE=$P run driver_p${P}_b timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --profile-steps 0
done
for PL in 1 3 0; do MV_PLANAR=$PL run tower_planar$PL $B; done
for PL in 1 0; do MV_PLANAR=$PL run empty_planar$PL $B --scenario Empty; done
for i in 1 2; do
  run hist_relaxed_$i $B
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_histfence.so run hist_fence_$i $B
done
