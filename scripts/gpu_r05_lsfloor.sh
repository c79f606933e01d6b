#!/bin/bash
# fp32 line-search slope floor (DMC_LS_SLOPE_ULPS) against the library without it: rate and parity of configs 2-5
mkdir -p gpurun_out
{
for c in 5 4 3 2; do for v in "" nofloor; do
  DMC_LIB_VARIANT=$v DMC_BENCH_NO_PMC=1 timeout 600 python bench.py --config $c --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $c', '${v:-main}', 'value %.5g ms %.4f rollout %.5g pipelined %.5g iter %.3f' % (d['value'], d['ms_per_step'], d['rollout']['value'], d.get('pipelined',{}).get('value',0), d['workload_stats']['mean_solver_iter']))
for k in ('open-loop','teacher-forced','teacher-forced-physics-step','f64-open-loop'):
  if k in p: print('     ', k, {a: ('%.3g' % b if isinstance(b, float) else b) for a, b in p[k].items()})
print('      warnings', d['warnings_after_run'], d.get('parity_error'))"
done; done
} > gpurun_out/lsfloor_ab.log 2>&1
cat gpurun_out/lsfloor_ab.log
