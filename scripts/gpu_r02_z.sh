#!/bin/bash
# Round-2 GPU session Z: lanes per environment revisited after the residency changes (cheetah 16 vs 32, humanoid 32 vs 64)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # label config args...
  local label=$1 c=$2; shift 2
  timeout 600 python bench.py --config $c --no-cpu-baseline --parity-steps 0 "$@" > gpurun_out/z_${label}_cfg$c.json 2> gpurun_out/z_${label}_cfg$c.err; echo "bench $label cfg $c rc=$?"; tail -2 gpurun_out/z_${label}_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/z_${label}_cfg$c.json'))
print('AB cfg$c $label', 'value', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'rollout', round(d['rollout']['value']), 'warn', sum(d['warnings_after_run']), {k: d['config']['info'].get(k) for k in ('lanes_per_env', 'waves_per_block', 'envs_per_block', 'envs_per_cu', 'grid', 'work_queue', 'static_id')})
PY
}
for rep in 1 2; do
  run l32 2 --lanes 32
  run l16 2 --lanes 16
  run l64 2 --lanes 64
done
for rep in 1 2; do
  run l64 3 --lanes 64
  run l32 3 --lanes 32
done
DMC_WAVES=2 python bench.py --config 2 --no-cpu-baseline --parity-steps 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg2 w2', round(d['value']), d['config']['info'])"
DMC_WAVES=1 python bench.py --config 2 --no-cpu-baseline --parity-steps 0 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg2 w1', round(d['value']), d['config']['info'])"
