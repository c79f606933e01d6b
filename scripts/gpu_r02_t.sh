#!/bin/bash
# Round-2 GPU session T: LDS diet 2 phase 2 (sparse M and the cold real tables in global memory too): full GPU
# test-suite, then phase-1 vs phase-2 library A/B of bench configs 4, 3, 5 on ONE box
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -rP > gpurun_out/pytest_gpu_t.log 2>&1; echo "pytest rc=$?"
grep -a "measured:\|environments above" gpurun_out/pytest_gpu_t.log; grep -a "passed\|failed" gpurun_out/pytest_gpu_t.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_t.log | head -20
for c in 4 3 5; do
  for v in p1 new p1 new; do
    if [ $v = new ]; then unset DMC_LIB_VARIANT; else export DMC_LIB_VARIANT=$v; fi
    timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/ab_${v}_cfg$c.json 2> gpurun_out/ab_${v}_cfg$c.err; echo "bench $v cfg $c rc=$?"
    python - <<PY
import json
d=json.load(open('gpurun_out/ab_${v}_cfg$c.json'))
print('AB cfg$c $v', 'value', round(d['value']), 'ms', d['ms_per_step'], 'rollout', round(d['rollout']['value']), 'warn', d['warnings_after_run'], {k: d['config']['info'][k] for k in ('waves_per_block', 'envs_per_block', 'envs_per_cu', 'lds_bytes_per_block', 'static_id', 'grid')})
PY
  done
done
unset DMC_LIB_VARIANT
for c in 4 3; do
timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/t_bench_cfg$c.json 2> gpurun_out/t_bench_cfg$c.err; echo "bench cfg$c rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/t_bench_cfg$c.json'))
print('cfg$c parity', {k: (v['max'], v['median']) for k, v in d.get('parity', {}).items() if isinstance(v, dict)}, d.get('parity_error'), 'warn', d['warnings_after_run'])
PY
done
