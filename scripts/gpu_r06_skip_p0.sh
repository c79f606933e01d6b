#!/bin/bash
# Round 6: the fp32 Newton line search without its evaluation at alpha = 0 (step_core.h primal_search, -DDMC_NO_LS_SKIP_P0
# restores it).  Plugin twins on one box, every config, parity legs included.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
export DMC_NO_STATIC=1 DMC_SPECIALISE=build
{
for rep in 1 2; do for c in ${CFGS:-5 2 3 4}; do
  for f in "" "-DDMC_NO_LS_SKIP_P0"; do
    DMC_SPEC_FLAGS="$f" DMC_BENCH_NO_PMC=1 timeout 400 python bench.py --config $c --no-cpu-baseline --parity-steps ${PSTEPS:-20} --pipeline 0 --extra 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity',{})
print('cfg $c rep $rep', repr('$f'), 'value %.5g ms %.4f rollout %.5g iter %.3f static %s' % (d['value'], d['ms_per_step'], d['rollout']['value'], d['workload_stats']['mean_solver_iter'], d.get('launch',{}).get('static_id')), {k: '%.3g' % v['max'] for k, v in p.items() if isinstance(v, dict) and 'max' in v})"
  done
done; done
} 2>&1 | tee gpurun_out/r06_skip_p0_ab.log
