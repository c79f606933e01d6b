#!/bin/bash
# round 5, session 4: final line-search policy (anchoring where it pays, exact-step floor for nv <= 16) -- A/B against the
# round-4 library with parity legs; specialise + device_env tests; the device environments' rates.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
show() { python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read()); p=d.get('parity') or {}
  tf=(p.get('teacher-forced-physics-step') or p.get('teacher-forced') or {}).get('per_step', {})
  te=(p.get('teacher-forced') or {}).get('per_step', {})
  print('$1', 'value %.5g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)), 'iters %.3f' % d.get('workload_stats',{}).get('mean_solver_iter',0), ('one physics step: median %.2e p99 %.2e max %.2e | one env-step: p99 %.2e max %.2e | open loop max %.2e' % (tf.get('median',0), tf.get('p99',0), tf.get('max',0), te.get('p99',0), te.get('max',0), (p.get('open-loop') or {}).get('max',0))) if tf else '')
except Exception as e: print('$1 FAILED', e)"; }
timeout 900 python -m pytest tests/test_specialise.py tests/test_device_env.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r05_spec_devenv_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_spec_devenv_tests.log
for cfg in 2 3 4 5; do
  DMC_LIB_VARIANT=r04 DMC_BENCH_NO_PMC=1 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | show "r04 cfg$cfg"
  DMC_BENCH_NO_PMC=1 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tee gpurun_out/r05_parity_final_cfg$cfg.json | show "new cfg$cfg"
done 2>&1 | tee gpurun_out/r05_ab_final_vs_round4.log
timeout 900 python scripts/device_env_runs.py > gpurun_out/r05_device_env_runs.log 2>&1; tail -3 gpurun_out/r05_device_env_runs.log
