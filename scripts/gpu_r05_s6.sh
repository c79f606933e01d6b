#!/bin/bash
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
for c in 5 4 3 2; do
  DMC_BENCH_NO_PMC=1 timeout 600 python bench.py --config $c --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $c', 'value %.5g ms %.4f rollout %.5g pipelined %.5g iter %.3f' % (d['value'], d['ms_per_step'], d['rollout']['value'], d.get('pipelined',{}).get('value',0), d['workload_stats']['mean_solver_iter']))
for k in ('open-loop','teacher-forced','teacher-forced-physics-step','f64-open-loop'):
  if k in p: print('     ', k, {a: ('%.3g' % b if isinstance(b, float) else b) for a, b in p[k].items() if a in ('max','median','p90','frac_le_1e4')}, (p[k].get('per_step') or {}).get('p99'))
print('      warnings', d['warnings_after_run'], d.get('parity_error'))"
done
} > gpurun_out/s6.log 2>&1
tail -40 gpurun_out/s6.log
