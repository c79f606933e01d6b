#!/bin/bash
# Round-2 GPU session AH: cache warm-up of the first environment during table staging vs the previous library
# (libdmc_hip_p10.so), config 2 on ONE box; full GPU tests; smoke
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/ah_${label}_cfg$c.json 2> gpurun_out/ah_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/ah_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ah_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']))
PY
}
for rep in 1 2 3; do
  run p10 2 DMC_LIB_VARIANT=p10
  run new 2 DMC_X=0
done
run p10 3 DMC_LIB_VARIANT=p10
run new 3 DMC_X=0
run p10 5 DMC_LIB_VARIANT=p10
run new 5 DMC_X=0
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_ae.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_ae.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_ae.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
