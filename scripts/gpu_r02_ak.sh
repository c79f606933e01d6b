#!/bin/bash
# Round-2 GPU session AK: dense M for the 27-dof humanoid (M v as a row product instead of the tree pass) vs the previous
# library (libdmc_hip_p12.so), config 3 on ONE box; full GPU tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/ak_${label}_cfg$c.json 2> gpurun_out/ak_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/ak_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ak_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']), d['config']['info']['envs_per_cu'])
PY
}
for rep in 1 2 3; do
  run p12 3 DMC_LIB_VARIANT=p12
  run new 3 DMC_X=0
done
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_ak.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_ak.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_ak.log | head -20
timeout 600 python bench.py --config 3 --no-cpu-baseline > gpurun_out/ak_bench_cfg3.json 2> gpurun_out/ak_bench_cfg3.err; echo "bench cfg3 rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/ak_bench_cfg3.json'))
print('cfg3 parity', {k: (v['max'], v['median']) for k, v in d.get('parity', {}).items() if isinstance(v, dict)}, d.get('parity_error'), 'warn', d['warnings_after_run'])
PY
