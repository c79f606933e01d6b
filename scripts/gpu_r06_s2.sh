#!/bin/bash
# Round 6, session 2: the GPU test tier on the sliced-queue library, smoke, the driver's bench command (with the extra
# legs), the 2-rank harness check of the overlapped exchange leg on one device (gloo: not a measurement).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rs -n 4 -x 2>&1 | tail -15 > gpurun_out/r06_s2_gputests.log; tail -4 gpurun_out/r06_s2_gputests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_s2_bench_driver_cmd.json 2> gpurun_out/r06_s2_bench_driver_cmd.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06_s2_bench_driver_cmd.json'))
print('cfg2', round(d['value']), d['ms_per_step'], 'frac', d['roofline']['frac'], 'cpu', d.get('cpu_baseline', {}).get('value'))
for k, v in d.get('extra', {}).get('configs', {}).items():
  print('extra cfg', k, {a: v.get(a) for a in ('value', 'kernel_ms_avg', 'leg_seconds', 'error')}, v.get('one_step_parity'))
print('env_step', d.get('extra', {}).get('env_step'))
PY
DMC_BENCH_SINGLE_DEVICE=1 DMC_BENCH_BACKEND=gloo DMC_BENCH_NO_PMC=1 timeout 600 python bench.py --gpus 2 --config 2 --steps 50 --warmup 5 --no-cpu-baseline --parity-steps 0 --pipeline 0 2> gpurun_out/r06_s2_2rank.err | tail -1 > gpurun_out/r06_s2_2rank_gloo_single_device.json
python -c "
import json; d=json.load(open('gpurun_out/r06_s2_2rank_gloo_single_device.json')); print('2-rank harness', d['n_gpus'], round(d['value']), json.dumps(d.get('collectives'))[:900])" || tail -5 gpurun_out/r06_s2_2rank.err
