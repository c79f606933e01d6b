#!/bin/bash
# Round-2 GPU session U: resident-only grid + device work queue (waves claim environments as they finish) vs the
# static one-workgroup-per-4-envs grid, same library, ONE box; workgroup shapes for config 4; then the full GPU tests
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config env...
  local label=$1 c=$2; shift 2
  env "$@" timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/u_${label}_cfg$c.json 2> gpurun_out/u_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"
  python - <<PY
import json
d=json.load(open('gpurun_out/u_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']), {k: d['config']['info'].get(k) for k in ('waves_per_block', 'envs_per_block', 'envs_per_cu', 'grid', 'work_queue')})
PY
}
for c in 4 3 5 2; do
  for rep in 1 2; do
    run static $c DMC_NO_QUEUE=1
    run queue $c DMC_X=0
  done
done
for w in 1 2 3; do
  run queue_w$w 4 DMC_WAVES=$w
  run static_w$w 4 DMC_WAVES=$w DMC_NO_QUEUE=1
done
for w in 1 2 3; do
  run queue_w$w 3 DMC_WAVES=$w
done
timeout 1500 python -m pytest tests -q -m gpu -rP > gpurun_out/pytest_gpu_u.log 2>&1; echo "pytest rc=$?"
grep -a "measured:\|environments above" gpurun_out/pytest_gpu_u.log; grep -a " passed\| failed" gpurun_out/pytest_gpu_u.log | tail -3; grep -a "^FAILED\|^ERROR" gpurun_out/pytest_gpu_u.log | head -20
