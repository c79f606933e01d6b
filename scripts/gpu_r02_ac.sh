#!/bin/bash
# Round-2 GPU session AC: 62 x 62 Cholesky with LDS column broadcasts vs v_readlane broadcasts (libdmc_hip_p6.so),
# config 4 on ONE box; GPU tests of the 62-dof models
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/ac_${label}_cfg$c.json 2> gpurun_out/ac_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/ac_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ac_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']))
PY
}
for rep in 1 2 3; do
  run p6 4 DMC_LIB_VARIANT=p6
  run new 4 DMC_X=0
done
timeout 1500 python -m pytest tests/test_gpu_suite.py tests/test_gpu_composer.py -q -m gpu > gpurun_out/pytest_gpu_ac.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_ac.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_ac.log | head -20
timeout 600 python bench.py --config 4 --no-cpu-baseline > gpurun_out/ac_bench_cfg4.json 2> gpurun_out/ac_bench_cfg4.err; echo "bench cfg4 rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/ac_bench_cfg4.json'))
print('cfg4 parity', {k: (v['max'], v['median']) for k, v in d.get('parity', {}).items() if isinstance(v, dict)}, d.get('parity_error'), 'warn', d['warnings_after_run'])
PY
