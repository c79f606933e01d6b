#!/bin/bash
# Round-2 GPU session AG: kinematic stash (poses / COM frame / velocities kept between legacy steps, self-validating)
# on vs off, same library, all four configs on ONE box; full GPU tests; smoke
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/ag_${label}_cfg$c.json 2> gpurun_out/ag_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/ag_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ag_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']))
PY
}
for c in 2 3 4 5; do
  for rep in 1 2; do
    run off $c DMC_NO_KSTASH=1
    run on $c DMC_X=0
  done
done
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_ag.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_ag.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_ag.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
timeout 600 python bench.py --config 2 --no-cpu-baseline > gpurun_out/ag_bench_cfg2.json 2> gpurun_out/ag_bench_cfg2.err; echo "bench cfg2 rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/ag_bench_cfg2.json'))
print('cfg2 parity', {k: (v['max'], v['median']) for k, v in d.get('parity', {}).items() if isinstance(v, dict)}, d.get('parity_error'), 'warn', d['warnings_after_run'])
PY
