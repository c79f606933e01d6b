#!/bin/bash
# soccer task layer as kernels: parity test, the composer GPU tests, env rate with / without (scripts/composer_runs.py)
cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_composer.py -x -q -m gpu -s 2>&1 | tail -15 | tee gpurun_out/r06_soccer_task_tests.log
ONLY=soccer GRAPH=1 T=300 timeout 600 python scripts/composer_runs.py > gpurun_out/r06_soccer_task_runs.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r06_soccer_task_runs.log'):
  if l.startswith('{'):
    r = json.loads(l); print(r['env'], r['B'], r['kwargs'], r['mode'], '%.1f k env-steps/s' % (r['env_steps_per_s'] / 1e3), 'episodes', r['episodes_ended'], 'warn', sum(r['warnings']))
PY
tail -3 gpurun_out/r06_soccer_task_runs.log | cut -c1-300
