#!/bin/bash
# Round-2 GPU session S: contact Jacobian rows + kept factor of M in the per-env global scratch (LDS diet 2):
# parity tests, then old-vs-new library A/B of bench configs 3, 4, 5 and 2 on ONE box
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_suite.py -x -q -m gpu -rP > gpurun_out/pytest_gpu_s.log 2>&1; echo "pytest rc=$?"
grep -a "measured:" gpurun_out/pytest_gpu_s.log; tail -5 gpurun_out/pytest_gpu_s.log
for c in 4 3 5 2; do
  for v in old new old new; do
    if [ $v = old ]; then export DMC_LIB_VARIANT=old; else unset DMC_LIB_VARIANT; fi
    timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/ab_${v}_cfg$c.json 2> gpurun_out/ab_${v}_cfg$c.err; echo "bench $v cfg $c rc=$?"
    python - <<PY
import json
d=json.load(open('gpurun_out/ab_${v}_cfg$c.json'))
print('AB cfg$c $v', 'value', round(d['value']), 'ms', d['ms_per_step'], 'rollout', round(d['rollout']['value']), 'warn', d['warnings_after_run'], {k: d['config']['info'][k] for k in ('waves_per_block', 'envs_per_block', 'envs_per_cu', 'lds_bytes_per_block', 'static_id', 'grid')})
PY
  done
done
unset DMC_LIB_VARIANT
timeout 600 python bench.py --config 4 --no-cpu-baseline > gpurun_out/s_bench_cfg4.json 2> gpurun_out/s_bench_cfg4.err; echo "bench cfg4 rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/s_bench_cfg4.json'))
print('cfg4 parity', d.get('parity'), d.get('parity_error'), 'warn', d['warnings_after_run'])
PY
