#!/bin/bash
# Round-2 GPU session L: tests, bench configs 3-5 after the geometry / cap changes, soak of every suite task
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
for c in 3 4 5 2; do
  timeout 900 python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_l_cfg$c.json 2> gpurun_out/bench_l_cfg$c.err; echo "bench cfg $c rc=$?"; tail -3 gpurun_out/bench_l_cfg$c.err
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_l_cfg$c.json'))
p=d.get('parity',{})
print('cfg$c value %.0f ms %.4f phys/s %.0f rollout %.0f warn %s'%(d['value'],d['ms_per_step'],d['physics_steps_per_s'],d['rollout']['value'],d['warnings_after_run']))
for k in ('open-loop','teacher-forced','f64-open-loop'):
  print('    ',k, {kk:(('%.2e'%vv) if isinstance(vv,float) else vv) for kk,vv in p.get(k,{}).items() if kk!='oracle_warnings'})
print('    info', d['config']['info'], d['workload_stats'])
PY
done
T=300 timeout 1500 python scripts/soak.py > gpurun_out/soak.log 2>&1; echo "soak rc=$?"; cut -c1-260 gpurun_out/soak.log | tail -50
