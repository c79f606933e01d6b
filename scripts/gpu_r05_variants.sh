#!/bin/bash
# A/B of on-demand specialised kernels built with different DMC_SPEC_FLAGS (scripts/spec_variants.py) on one box:
#   CFG=2 REPS=2 bash scripts/gpu_r05_variants.sh "" "-fno-slp-vectorize" ...
mkdir -p gpurun_out
CFG=${CFG:-2}; REPS=${REPS:-2}
{
for rep in $(seq $REPS); do for f in "$@"; do
  DMC_NO_STATIC=1 DMC_SPEC_FLAGS="$f" DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $CFG --no-cpu-baseline --parity-steps ${PARITY_STEPS:-0} --pipeline 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $CFG', repr('$f'), 'value %.5g ms %.4f rollout %.5g iter %.3f static_id %s' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter'], d.get('launch',{}).get('static_id')), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v})"
done; done
} > gpurun_out/variants.log 2>&1
cat gpurun_out/variants.log
