#!/bin/bash
# Round-2 GPU session AP: bench lines of the four configs with the final library (profiles/r02_bench_cfg*.json)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out gpurun_out/profiles_new
export TMPDIR=/tmp
for c in 2 3 4 5; do
  timeout 300 python bench.py --config $c --cpu-seconds 4 > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err; echo "bench cfg $c rc=$?"
  cp gpurun_out/bench_cfg$c.json gpurun_out/profiles_new/r02_bench_cfg$c.json
  python -c "import json; d=json.load(open('gpurun_out/bench_cfg$c.json')); print('cfg$c', round(d['value']), d['ms_per_step'], round(d['rollout']['value']), d['cpu_baseline']['value'], {k: v['max'] for k, v in d['parity'].items() if isinstance(v, dict)}, sum(d['warnings_after_run']))"
done
