#!/bin/bash
# Round-2 GPU session X: all-dense constraint rows for the small models (nv <= 16) vs the previous library
# (libdmc_hip_p3.so), config 2 on ONE box; full GPU tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/x_${label}_cfg$c.json 2> gpurun_out/x_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/x_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/x_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']), {k: d['config']['info'].get(k) for k in ('waves_per_block', 'envs_per_block', 'envs_per_cu', 'grid', 'work_queue', 'lds_bytes_per_block')})
PY
}
for rep in 1 2 3; do
  run p3 2 DMC_LIB_VARIANT=p3
  run new 2 DMC_X=0
done
run p3 3 DMC_LIB_VARIANT=p3
run new 3 DMC_X=0
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_x.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_x.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_x.log | head -20
