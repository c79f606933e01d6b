#!/bin/bash
# A/B of named plugin objects (dm_control_amd/_spec_cache/v_<tag>_cfg<N>.so, scripts/spec_variants.py NAME=<tag>) on one box:
#   CFG=2 REPS=2 bash scripts/gpu_r05_plugins.sh base new ...     PARITY_STEPS=<n> adds the parity legs
mkdir -p gpurun_out
CFG=${CFG:-2}; REPS=${REPS:-2}
{
for rep in $(seq $REPS); do for t in "$@"; do
  DMC_NO_STATIC=1 DMC_SPEC_PLUGIN=$PWD/dm_control_amd/_spec_cache/v_${t}_cfg${CFG}.so DMC_BENCH_NO_PMC=1 timeout 300 python bench.py --config $CFG --no-cpu-baseline --parity-steps ${PARITY_STEPS:-0} --pipeline 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $CFG %-12s' % '$t', 'value %.5g ms %.4f rollout %.5g iter %.3f' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter']), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v})"
done; done
} > gpurun_out/plugins_cfg$CFG.log 2>&1
cat gpurun_out/plugins_cfg$CFG.log
