#!/bin/bash
# Round-2 GPU session W: noslip blocks with one global round trip (rows of A in registers) vs the previous library
# (libdmc_hip_p3.so), configs 5 and 4 on ONE box; noslip / soccer / CMU parity tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/w_${label}_cfg$c.json 2> gpurun_out/w_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/w_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/w_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']), {k: d['config']['info'].get(k) for k in ('waves_per_block', 'envs_per_block', 'envs_per_cu', 'grid', 'work_queue')})
PY
}
for c in 5 4; do
  for rep in 1 2; do
    run p3 $c DMC_LIB_VARIANT=p3
    run new $c DMC_X=0
  done
done
timeout 1500 python -m pytest tests/test_gpu_suite.py tests/test_gpu_parity.py tests/test_gpu_composer.py -q -m gpu > gpurun_out/pytest_gpu_w.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_w.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_w.log | head -20
for c in 5 4; do
timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/w_bench_cfg$c.json 2> gpurun_out/w_bench_cfg$c.err; echo "bench cfg$c rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/w_bench_cfg$c.json'))
print('cfg$c parity', {k: (v['max'], v['median']) for k, v in d.get('parity', {}).items() if isinstance(v, dict)}, d.get('parity_error'), 'warn', d['warnings_after_run'])
PY
done
