#!/bin/bash
# Round-2 GPU session AA: fp32 Cholesky pivots through v_rsq_f32 vs the previous library (libdmc_hip_p4.so), all four
# configs on ONE box; full GPU tests; the all-tasks soak
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/aa_${label}_cfg$c.json 2> gpurun_out/aa_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/aa_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/aa_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']))
PY
}
for c in 4 3 2 5; do
  for rep in 1 2; do
    run p4 $c DMC_LIB_VARIANT=p4
    run new $c DMC_X=0
  done
done
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_aa.log 2>&1; echo "pytest rc=$?"
grep -a " passed\| failed" gpurun_out/pytest_gpu_aa.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_aa.log | head -20
T=300 timeout 1500 python scripts/soak.py > gpurun_out/soak.log 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/soak.log | cut -c1-300
python - <<PY
import json
d=json.load(open('gpurun_out/soak.json'))
print('soak tasks', len(d), 'errors', [r for r in d if 'error' in r][:3], 'nonzero warnings', [(r['task'], r['warnings']) for r in d if 'warnings' in r and sum(r['warnings'])])
PY
