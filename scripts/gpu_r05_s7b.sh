#!/bin/bash
# session 7: A/B of the working library against the library of commit 5ccff94 (libdmc_hip_base.so) on one box
# CFGS="2 3" REPS=2 VARIANTS="'' base"
mkdir -p gpurun_out
CFGS=${CFGS:-2}; REPS=${REPS:-2}
{
for c in $CFGS; do for rep in $(seq $REPS); do for v in "" base; do
  DMC_LIB_VARIANT=$v DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $c --no-cpu-baseline --parity-steps 0 --pipeline 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg $c', '${v:-main}', 'value %.5g ms %.4f rollout %.5g iter %.3f' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter']))"
done; done; done
if [ -n "$PARITY" ]; then for c in $CFGS; do
  DMC_BENCH_NO_PMC=1 timeout 600 python bench.py --config $c --no-cpu-baseline --pipeline 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $c parity (main)')
for k in ('open-loop','teacher-forced','teacher-forced-physics-step','f64-open-loop'):
  if k in p: print('     ', k, {a: ('%.3g' % b if isinstance(b, float) else b) for a, b in p[k].items() if a in ('max','median','p90','frac_le_1e4')})
print('      warnings', d['warnings_after_run'], d.get('parity_error'))"
done; fi
if [ -n "$TESTS" ]; then timeout 1200 python -m pytest tests -q -m gpu -x $TESTS 2>&1 | tail -5; fi
} > gpurun_out/s7b.log 2>&1
cat gpurun_out/s7b.log
