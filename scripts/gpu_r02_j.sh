#!/bin/bash
# Round-2 GPU session J: full GPU test-suite, bench configs 3-5 with the new caps, fp64 fit of the 62-dof models
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log
python - <<'PY'
import sys; sys.path.insert(0, '.')
from dm_control_amd import mjcf_compiler as mc
from dm_control_amd.batch import BatchedPhysics
from dm_control_amd.suite import common
for name, ncon in (('cmu_2019_position_floor', 48), ('cmu_2019_position_floor', 32), ('humanoid_CMU', 48), ('soccer_2v2_boxhead', 24)):
  m = mc.compile_xml(common.read_model(name + '.xml'))
  try:
    b = BatchedPhysics(m, 64, precision=64, nconmax=ncon)
    b.step(3); b.sync()
    print('fp64', name, ncon, b.info()); b.close()
  except Exception as ex:
    print('fp64', name, ncon, 'FAILED', repr(ex)[:200])
PY
for c in 4 5 3; do
  timeout 900 python bench.py --config $c > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err; echo "bench cfg $c rc=$?"; tail -3 gpurun_out/bench_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_cfg$c.json'))
print('cfg$c', 'value', d['value'], 'ms', d['ms_per_step'], 'phys/s', d['physics_steps_per_s'], 'rollout', d['rollout']['value'], 'parity', d.get('parity'), d.get('parity_error'), 'cpu', d.get('cpu_baseline'), 'warn', d['warnings_after_run'], d['workload_stats'], 'frac', d['roofline']['frac'], d['config']['info'])
PY
done
