#!/bin/bash
# Round-4 opening measurements with the round-3 kernel (+ implicitfast / profiling changes): device environments
# (eager + graph), quick bench lines of configs 3-5.  Outputs -> gpurun_out/r04_base_*
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
GRAPH=1 T=300 timeout 400 python scripts/composer_runs.py > gpurun_out/r04_base_composer.log 2>&1; echo "composer rc=$?"
cp gpurun_out/composer_runs.json gpurun_out/r04_base_composer_runs.json
for c in 5 4 3; do
  timeout 200 python bench.py --config $c --no-cpu-baseline --parity-steps 0 > gpurun_out/r04_base_bench_cfg$c.json 2> gpurun_out/r04_base_bench_cfg$c.err; echo "bench cfg$c rc=$?"
done
python - <<'PY'
import json
for r in json.load(open('gpurun_out/composer_runs.json')):
  print(r['env'], r['B'], r['kwargs'], r['mode'], round(r['env_steps_per_s']))
for c in (5, 4, 3):
  d = json.load(open('gpurun_out/r04_base_bench_cfg%d.json' % c)); print('cfg', c, round(d['value']), d['ms_per_step'], d['rollout']['value'])
PY
