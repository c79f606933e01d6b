#!/bin/bash
# Round-2 GPU session V: longest-first scheduling of queued launches (items ordered by their cost in the previous
# launch) vs index order, same library, ONE box; then the full GPU tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/v_${label}_cfg$c.json 2> gpurun_out/v_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/v_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/v_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']), {k: d['config']['info'].get(k) for k in ('waves_per_block', 'envs_per_block', 'envs_per_cu', 'grid', 'work_queue')})
PY
}
for c in 4 3; do
  for rep in 1 2; do
    run index $c DMC_NO_LPT=1
    run lpt $c DMC_X=0
  done
done
run static 4 DMC_NO_QUEUE=1
timeout 1500 python -m pytest tests -q -m gpu -rP > gpurun_out/pytest_gpu_v.log 2>&1; echo "pytest rc=$?"
grep -a "measured:\|environments above" gpurun_out/pytest_gpu_v.log; grep -a " passed\| failed" gpurun_out/pytest_gpu_v.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_v.log | head -20
