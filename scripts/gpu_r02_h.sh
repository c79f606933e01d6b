#!/bin/bash
# Round-2 GPU session H: new tests; r01 vs now on the same box (cheetah phases + bench); bench configs 2-5
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step1 or stash" > gpurun_out/pytest_gpu_h.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu_h.log
(cd _r01 && mkdir -p gpurun_out && timeout 300 python bench.py --no-cpu-baseline > $R/gpurun_out/bench_r01lib.json 2> $R/gpurun_out/bench_r01lib.err; echo "r01 bench rc=$?"; MODEL=cheetah NSUB=1 B=4096 timeout 300 python scripts/phase_profile_model.py > $R/gpurun_out/phase_cheetah_r01lib.log 2>&1; cat $R/gpurun_out/phase_cheetah_r01lib.log | cut -c1-200)
python -c "
import json; d=json.load(open('gpurun_out/bench_r01lib.json')); print('r01lib', d['value'], d['ms_per_step'], d['rollout']['value'])"
MODEL=cheetah NSUB=1 B=4096 timeout 300 python scripts/phase_profile_model.py > gpurun_out/phase_cheetah.log 2>&1; cat gpurun_out/phase_cheetah.log | cut -c1-200
for c in 2 3 4 5; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err; echo "bench cfg $c rc=$?"; tail -3 gpurun_out/bench_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_cfg$c.json'))
print('cfg$c', 'value', d['value'], 'ms', d['ms_per_step'], 'phys/s', d['physics_steps_per_s'], 'rollout', d['rollout']['value'], 'parity', d.get('parity'), d.get('parity_error'), 'cpu', d.get('cpu_baseline'), 'warn', d['warnings_after_run'], d['workload_stats'], 'frac', d['roofline']['frac'])
PY
done
