#!/bin/bash
# round 5, session 3: (1) the line-search / floor variants: speed and parity A/B on one box; (2) device_env on the device:
# 45 tasks vs the host task + env-steps/s next to physics-only; (3) on-demand specialisation: tests, and a plugin twin of
# each config model (DMC_NO_STATIC=1 hides the baked kernel, the cache holds the plugin) against its baked kernel.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
show() { python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read()); p=d.get('parity') or {}
  tf=(p.get('teacher-forced-physics-step') or p.get('teacher-forced') or {}).get('per_step', {})
  print('$1', 'value %.5g ms %.5f rollout %.4g' % (d['value'], d['ms_per_step'], d.get('rollout',{}).get('value',0)), 'iters %.3f' % d.get('workload_stats',{}).get('mean_solver_iter',0), 'one-step p99 %.2e max %.2e' % (tf.get('p99',0), tf.get('max',0)) if tf else '')
except Exception as e: print('$1 FAILED', e)"; }
timeout 900 python -m pytest tests/test_specialise.py tests/test_device_env.py -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/r05_spec_devenv_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_spec_devenv_tests.log
for cfg in 3 4 5; do
  for v in main nofloor noanchor; do
    if [ $v = main ]; then unset DMC_LIB_VARIANT; else export DMC_LIB_VARIANT=$v; fi
    DMC_BENCH_NO_PMC=1 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | show "$v cfg$cfg"
  done
done 2>&1 | tee gpurun_out/r05_ab_variants.log
unset DMC_LIB_VARIANT
for cfg in 2 3 4 5; do
  python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "baked cfg$cfg"
  DMC_NO_STATIC=1 DMC_SPECIALISE=cached python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "plugin cfg$cfg"
  DMC_NO_STATIC=1 DMC_SPECIALISE=0 python bench.py --config $cfg --no-cpu-baseline --parity-steps 0 2>/dev/null | show "generic cfg$cfg"
done 2>&1 | tee gpurun_out/r05_generic_vs_static.log
timeout 900 python scripts/device_env_runs.py > gpurun_out/r05_device_env_runs.log 2>&1; tail -4 gpurun_out/r05_device_env_runs.log
