#!/bin/bash
# Round 6: suite/fused_env.py on the device -- tests (stand-alone task kernels for all 45 tasks, the task layer inside
# the step kernel), then env rate vs physics-only rate for all 45 tasks with the in-kernel task layer (the (model, task)
# kernels come prebuilt in the in-tree cache: scripts/prebuild_task_kernels.py).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_env.py tests/test_native_abi.py tests/test_specialise.py tests/test_device_env.py -m gpu -q -n 4 2>&1 | tail -12
timeout 1500 python scripts/fused_env_runs.py > gpurun_out/r06_fused_env_runs.log 2>&1; tail -1 gpurun_out/r06_fused_env_runs.log
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_fused_env_runs.json'))
for r in d['runs']:
  if 'error' in r: print(r)
  else: print('%-28s ratio %.3f  env %8.1f us  physics %8.1f us  make %.1f s  %s' % (r['task'], r['ratio'], r['us_per_step'], r['us_physics'], r['make_seconds'], r['warnings']))
PY
